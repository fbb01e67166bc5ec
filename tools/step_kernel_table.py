#!/usr/bin/env python3
"""Markdown table of ONE training step's kernels from a tools/step_timeline.py listing (the step between two SGD
kernels of a rocprofv3 kernel trace): launches, total and average time per kernel -- the whole-run `--stats` summary
also counts start-up work (frozen-affine calibration, first-call transforms).
    python tools/step_kernel_table.py gpurun_out/TAG_timeline.txt [rows]"""
import collections
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "at::", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:80]


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    n, tot = collections.Counter(), collections.Counter()
    span = ""
    for line in open(path):
        m = re.match(r"\s*([\d.]+)\s+gap\s+(-?[\d.]+)\s+dur\s+([\d.]+)\s+(.*)", line)
        if m:
            k = short(m.group(4))
            n[k] += 1
            tot[k] += float(m.group(3))
        elif line.startswith("#"):
            span = line[1:].strip()
    total = sum(tot.values())
    conv = sum(v for k, v in tot.items() if re.search(r"igemm|wgrad", k))
    print("One step: %s.  Sum of kernel times %.2f ms (two streams overlap), %d launches." % (span, total / 1e3, sum(n.values())))
    print("Conv kernels (igemm / igemm3x3 / wgrad): %.2f ms = %.0f %% of the kernel time; everything else %.2f ms.\n"
          % (conv / 1e3, 100 * conv / total, (total - conv) / 1e3))
    print("| kernel | launches | ms | avg us | % |\n|---|---|---|---|---|")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:top]:
        print("| `%s` | %d | %.3f | %.1f | %.1f |" % (k, n[k], v / 1e3, v / n[k], 100 * v / total))


if __name__ == "__main__":
    main()
