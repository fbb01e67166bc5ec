#!/usr/bin/env python3
"""Side-by-side table of tools/bench_conv.py outputs concatenated into one file with '== <label>' separators.
    python tools/ab_table.py FILE [baseline-label]"""
import re
import sys

txt = open(sys.argv[1]).read()
secs = re.split(r'== (.*)\n', txt)
data, order = {}, []
for i in range(1, len(secs), 2):
    key = secs[i].strip()
    order.append(key)
    rows = {}
    for line in secs[i + 1].splitlines():
        m = re.match(r'(\S+)\s+([\d.]+) \|\s+([\d.]+)\s+([\d.]+) \|\s+([\d.]+)\s+([\d.]+) \|\s+([\d.]+)', line)
        if m:
            rows[m.group(1)] = (float(m.group(3)), float(m.group(5)), float(m.group(7)))
    data[key] = rows
base = sys.argv[2] if len(sys.argv) > 2 else order[0]
print("baseline:", base)
print("%-22s" % "layer" + "".join("%34s" % k[:32] for k in order if k != base))
for name in data[base]:
    line = "%-22s" % name
    for k in order:
        if k == base or name not in data[k]:
            continue
        a, b = data[base][name], data[k][name]
        line += "   f %6.1f>%6.1f(%+4.0f%%) d %6.1f>%6.1f(%+4.0f%%)"[:0] + "  f%6.1f>%6.1f %+4.0f%% d%6.1f>%6.1f %+4.0f%%" % (
            a[0], b[0], 100 * (b[0] / a[0] - 1), a[1], b[1], 100 * (b[1] / a[1] - 1))
    print(line)
