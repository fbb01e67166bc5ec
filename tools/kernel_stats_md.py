#!/usr/bin/env python3
"""Markdown table (per training step) from a rocprofv3 --kernel-trace --stats kernel_stats.csv.

    python tools/kernel_stats_md.py profiles/round1_kernel_stats.csv 13 [rows]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:80]


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
    calls = sum(float(r["Calls"]) for r in rows)
    conv = sum(float(r["TotalDurationNs"]) for r in rows if re.search(r"igemm|wgrad", r["Name"])) / 1e6
    print("Total kernel time %.1f ms = %.2f ms per step over %g steps; %d kernel launches per step." %
          (total, total / steps, steps, round(calls / steps)))
    print("Conv kernels (igemm / igemm3x3 / wgrad): %.2f ms/step = %.0f %% of GPU time; everything else %.2f ms/step.\n"
          % (conv / steps, 100 * conv / total, (total - conv) / steps))
    print("| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        t = float(r["TotalDurationNs"]) / 1e6
        print("| `%s` | %.1f | %.3f | %.1f | %.1f |" % (short(r["Name"]), float(r["Calls"]) / steps, t / steps,
                                                       float(r["AverageNs"]) / 1e3, 100 * t / total))


if __name__ == "__main__":
    main()
