#!/usr/bin/env python3
"""Summary of tools/pmc_traffic.sh: per layer and direction, the memory-side bytes of every kernel the call launches
(FETCH_SIZE x 2 on gfx950, WRITE_SIZE as counted: MI355X_MICROARCH.md) next to the call's algorithmic bytes."""
import collections
import csv
import glob
import re
import sys

sys.path.insert(0, "tools")
from bench_conv import LAYERS  # noqa: E402


def alg_bytes(name, d):
    for nm, N, C, H, W, K, R, st, pad, g, _ in LAYERS:
        if nm == name:
            P, Q = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
            x, y, w = 4.0 * N * C * H * W, 4.0 * N * K * P * Q, 4.0 * K * R * R * (C // g)
            if d == "fwd":
                return x + y + w + y, "x + y + w + residual"           # --epi res: affine + residual + ReLU
            if d == "dgrad":
                return y + w + 2 * x, "dy + w + dx read (accumulate) + dx"
            return x + y + w, "x + dy + dw"
    return 0.0, "?"


def load(tag, layer, d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("gpurun_out/%s_tr/%s_%s_%s/**/*counter_collection.csv" % (tag, layer, d, counter), recursive=True):
        for r in csv.DictReader(open(f)):
            nm = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:44]
            if not re.search(r"igemm|wgrad|splitk|seed_rows|weight_to_dgrad|fillBuffer|epilogue", nm):
                continue
            if r["Counter_Name"] != counter:
                continue
            a = agg[nm]
            a[0] += 1
            a[1] += float(r["Counter_Value"]) * 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
    return agg


def main():
    tag, layers = sys.argv[1], sys.argv[2].split()
    calls = 5                                                   # --iters 4 + one warm-up
    print("%-18s %-6s %-46s %6s %9s %9s | %9s %6s" % ("layer", "dir", "kernel", "n/call", "fetch MB", "write MB",
                                                      "alg MB", "ratio"))
    for L in layers:
        for d in ("fwd", "dgrad", "wgrad"):
            fe, wr = load(tag, L, d, "FETCH_SIZE"), load(tag, L, d, "WRITE_SIZE")
            alg, what = alg_bytes(L, d)
            tot = 0.0
            for nm in sorted(set(fe) | set(wr)):
                n = max(fe.get(nm, [0])[0], wr.get(nm, [0])[0])
                f = fe.get(nm, [0, 0.0])[1] / calls
                w = wr.get(nm, [0, 0.0])[1] / calls
                tot += f + w
                print("%-18s %-6s %-46s %6.1f %9.1f %9.1f |" % (L, d, nm, n / calls, f / 1e6, w / 1e6))
            print("%-18s %-6s %-46s %6s %9s %9.1f | %9.1f %6.2f   (%s)" % (L, d, "= all kernels of the call", "", "",
                                                                          tot / 1e6, alg / 1e6, tot / alg if alg else 0, what))


if __name__ == "__main__":
    main()
