#!/usr/bin/env python3
"""In-process sweep of the igemm planner's choice (CPM_IGEMM_FORCE="bm,bn,split" is read per call), bf16x3, forward and
data gradient with pre-split weights:  python tools/sweep_igemm.py [--filter a,b]
Prints us per (tile, split) and the planner's own choice (force "")."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from bench_conv import LAYERS, timeit  # noqa: E402
from pet.lib.ops import _hip, conv as ops  # noqa: E402

CL = torch.channels_last
TILES = [(128, 128), (128, 64), (64, 64)]
SPLITS = [1, 2, 3, 4, 6, 8]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    _hip.set_conv_math("bf16x3")
    for name, N, C, H, W, K, R, st, pad, g, cnt in LAYERS:
        if a.filter and not any(f in name for f in a.filter.split(",")):
            continue
        x = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=CL)
        w = (torch.randn(K, C // g, R, R, device="cuda") * 0.05).contiguous(memory_format=CL)
        w4 = ops.split_w4(w)
        P, Q = ops.out_size(H, R, st, pad), ops.out_size(W, R, st, pad)
        dy = torch.randn(N, K, P, Q, device="cuda").contiguous(memory_format=CL)
        fwd = lambda: ops.conv2d_forward(x, w, None, None, None, 0, False, st, pad, 1, g, w4=w4)
        dgr = lambda: ops.conv2d_backward_data(dy, w, (N, C, H, W), st, pad, 1, g)
        for what, fn in (("fwd", fwd), ("dgrad", dgr)):
            os.environ.pop("CPM_IGEMM_FORCE", None)
            row = ["plan:%.1f" % (timeit(fn, 10) * 1e3)]
            for bm, bn in TILES:
                for sp in SPLITS:
                    if sp > 1 and R * R * (C // g) < 64 * sp:
                        continue
                    os.environ["CPM_IGEMM_FORCE"] = "%d,%d,%d" % (bm, bn, sp)
                    try:
                        row.append("%dx%d/%d:%.1f" % (bm, bn, sp, timeit(fn, 10) * 1e3))
                    except RuntimeError as e:
                        row.append("%dx%d/%d:ERR" % (bm, bn, sp))
            os.environ.pop("CPM_IGEMM_FORCE", None)
            print("%-18s %-5s | %s" % (name, what, " ".join(row)), flush=True)


if __name__ == "__main__":
    main()
