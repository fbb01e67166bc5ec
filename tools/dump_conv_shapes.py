"""Per-shape conv time of a training step from the library's per-launch log (CPM_PROF_DUMP), e.g. to compare forced
tile choices (CPM_IGEMM_FORCE) on the step's real epilogues.  usage: dump_conv_shapes.py a.csv [b.csv ...]"""
import collections
import csv
import sys


def load(path, steps=2):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        key = (r["kind"], r["N"], r["H"], r["W"], r["C"], r["K"], r["R"], r["stride"], r["groups"])
        a = agg[key]
        a[0] += 1
        a[1] += float(r["gflop"])
        a[2] += float(r["ms"])
    return {k: (v[0] / steps, v[2] / steps, v[1] / max(v[2], 1e-9)) for k, v in agg.items()}


def main():
    tabs = [load(p) for p in sys.argv[1:]]
    keys = sorted(tabs[0], key=lambda k: -tabs[0][k][1])[:60]
    kinds = {"0": "fwd", "1": "dgrad", "2": "wgrad"}
    for k in keys:
        cols = ["%7.3f ms %6.1f TF" % (t[k][1], t[k][2]) if k in t else "   -   " for t in tabs]
        print("%-5s %-34s x%-4.1f %s" % (kinds[k[0]], " ".join(k[1:]), tabs[0][k][0], " ".join(cols)))
    print("total", ["%.2f" % sum(v[1] for v in t.values()) for t in tabs])


if __name__ == "__main__":
    main()
