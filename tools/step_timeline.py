#!/usr/bin/env python3
"""Kernel-by-kernel timeline of one training step from a rocprofv3 --kernel-trace CSV: start time (ms after the
previous step's SGD kernel), idle gap before the kernel (us), duration (us), name.
    python tools/step_timeline.py kernel_trace.csv [step_index] > timeline.txt"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:100]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    sgd = [i for i, e in enumerate(ev) if "sgd_kernel" in e[2]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(sgd) - 3
    step = ev[sgd[k]:sgd[k + 1] + 1]
    # times count from the START of the previous step's SGD kernel (it may run beside this step's frozen layers:
    # FlatSGD.overlap_next_forward); gaps are device idle time (no kernel of any stream running)
    t0 = step[0][0]
    end = step[0][1]
    busy = idle = 0.0
    print("%8.3f  gap %6.1f  dur %7.1f  %s" % (0.0, 0.0, (step[0][1] - step[0][0]) / 1e3, short(step[0][2])))
    for s, e, n in step[1:]:
        gap = (s - end) / 1e3
        idle += max(gap, 0.0)
        busy += (e - s) / 1e3
        print("%8.3f  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e6, gap, (e - s) / 1e3, short(n)))
        end = max(end, e)
    print("# %d kernels, busy %.2f ms, idle %.2f ms, span %.2f ms" % (len(step) - 1, busy / 1e3, idle / 1e3,
                                                                      (end - t0) / 1e6))


if __name__ == "__main__":
    main()
