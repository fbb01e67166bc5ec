"""Idle gaps of the GPU inside one training step, from a rocprofv3 --kernel-trace CSV: the largest pauses between
consecutive kernels (name before / after), i.e. where the host (sync round trips, Python launch path) holds the
device up.  usage: step_gaps.py kernel_trace.csv [which_step]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    sgd = [i for i, e in enumerate(ev) if "sgd_kernel" in e[2]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(sgd) - 2
    lo, hi = sgd[k], sgd[k + 1]
    step = ev[lo:hi + 1]
    span = (step[-1][1] - step[0][1]) / 1e6
    busy = sum(e[1] - e[0] for e in step[1:]) / 1e6
    print("step %d: %.2f ms between two SGD kernels, GPU busy %.2f ms, %d kernels" % (k, span, busy, len(step) - 1))
    gaps = []
    end = step[0][1]
    for i in range(1, len(step)):
        s, e, n = step[i]
        if s > end:
            gaps.append(((s - end) / 1e3, step[i - 1][2][:60], n[:60], (s - step[0][1]) / 1e6))
        end = max(end, e)
    tot = sum(g[0] for g in gaps) / 1e3
    print("idle %.2f ms in %d gaps; gaps > 20 us: %.2f ms" % (tot, len(gaps), sum(g[0] for g in gaps if g[0] > 20) / 1e3))
    for g in sorted(gaps, reverse=True)[:25]:
        print("%8.1f us at t=%6.2f ms  after %-60s before %s" % (g[0], g[3], g[1], g[2]))


if __name__ == "__main__":
    main()
