#!/usr/bin/env python3
"""In-process sweep of the weight gradient's pixel split (CPM_WGRAD_SPLIT is read per call) for one CPM_WGRAD_KS:
    CPM_WGRAD_KS=2 python tools/sweep_wgrad.py [--filter a,b]
Prints us per (layer, split); split 0 = the planner's choice."""
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
SPLITS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    _hip.set_conv_math("bf16x3")
    print("ks", os.environ.get("CPM_WGRAD_KS", "1"))
    for name, N, C, H, W, K, R, st, pad, g, cnt in LAYERS:
        if a.filter and not any(f in name for f in a.filter.split(",")):
            continue
        x = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=CL)
        w = (torch.randn(K, C // g, R, R, device="cuda") * 0.05).contiguous(memory_format=CL)
        P, Q = ops.out_size(H, R, st, pad), ops.out_size(W, R, st, pad)
        dy = torch.randn(N, K, P, Q, device="cuda").contiguous(memory_format=CL)
        dw = torch.zeros_like(w)
        chunks = (N * P * Q + 31) // 32
        tiles = ((K // g + 127) // 128) * ((C // g + 127) // 128) * R * R * g
        row = []
        for s in SPLITS:
            if s > chunks // 4 and s:
                continue
            if s:
                os.environ["CPM_WGRAD_SPLIT"] = str(s)
            else:
                os.environ.pop("CPM_WGRAD_SPLIT", None)
            t = timeit(lambda: ops.conv2d_backward_weight(x, dy, w, st, pad, 1, g, out=dw), 10)
            row.append("%d:%.1f" % (s, t * 1e3))
        os.environ.pop("CPM_WGRAD_SPLIT", None)
        print("%-18s chunks %5d tiles %4d | %s" % (name, chunks, tiles, " ".join(row)), flush=True)


if __name__ == "__main__":
    main()
