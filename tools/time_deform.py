#!/usr/bin/env python3
"""Times of the deformable 3x3 of X-101-64x4d-FPN-DCN's stages at bs=1, 800x1333 through the C-ABI: the fused kernels
(forward, data + offset gradient, weight gradient) beside the column-matrix path's pieces.
    python tools/time_deform.py [--zero-offsets]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))

SHAPES = [(256, 200, 336, False), (512, 100, 168, True), (1024, 50, 84, True), (2048, 25, 42, True)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--zero-offsets", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--dy-scale", type=float, default=1.0)
    ap.add_argument("--offset-range", type=float, default=1.5, help="offsets uniform in +-R, independent per pixel")
    a = ap.parse_args()
    import pet.lib.ops  # noqa: F401
    from pet.lib.ops import _hip as H
    from pet.lib.ops import conv as F
    dc = sys.modules["pet.lib.ops.deform_conv"]
    H.set_conv_math("bf16x3")
    L = H.lib()
    CL = torch.channels_last
    for C, Hh, W, deform in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(1, C, Hh, W, generator=g).cuda().contiguous(memory_format=CL)
        w = (torch.randn(C, C // 64, 3, 3, generator=g) * 0.1).cuda().contiguous(memory_format=CL)
        off = None
        if deform:
            off = torch.zeros(1, 18, Hh, W) if a.zero_offsets else (torch.rand(1, 18, Hh, W, generator=g) * 2 - 1) * a.offset_range
            off = off.cuda().contiguous(memory_format=CL)
        dy = (torch.randn(1, C, Hh, W, generator=g).double() * a.dy_scale).float().cuda().contiguous(memory_format=CL)
        geom = dc._geom(x.shape, w.shape, 1, 1, 1, 64, 1)
        fa = dc._fused_args(geom, C)
        y = torch.empty_like(x)
        dx = torch.zeros_like(x)
        doff = torch.empty_like(off) if deform else None
        dw = torch.zeros_like(w)
        s = H.stream()
        out = {}
        if not a.only or "fused" in a.only:
            out["fwd"] = timed(lambda: L.cpm_deform_conv_forward(H.ptr(x), H.ptr(off), H.ptr(w), None, None, 0, *fa,
                                                                 H.ptr(y), s))
            out["bwd_dx"] = timed(lambda: L.cpm_deform_conv_backward_data(H.ptr(dy), H.ptr(off), H.ptr(w), *fa,
                                                                          H.ptr(dx), s))
            out["dw"] = timed(lambda: L.cpm_deform_conv_backward_params(H.ptr(dy), H.ptr(x), H.ptr(off), H.ptr(w), *fa,
                                                                        H.ptr(dw), None, s))
            if deform:
                out["dw+doff"] = timed(lambda: L.cpm_deform_conv_backward_params(
                    H.ptr(dy), H.ptr(x), H.ptr(off), H.ptr(w), *fa, H.ptr(dw), H.ptr(doff), s))
                out["doff"] = timed(lambda: L.cpm_deform_conv_backward_params(
                    H.ptr(dy), H.ptr(x), H.ptr(off), H.ptr(w), *fa, None, H.ptr(doff), s))
        if not a.only or "cols" in a.only:
            cols = dc.sample_columns(x, off, geom)
            w1 = dc._w1x1(w)
            out["im2col"] = timed(lambda: dc.sample_columns(x, off, geom))
            out["gemm"] = timed(lambda: F.conv2d_forward(cols, w1, None, None, None, 0, False, 1, 0, 1, 64))
            dcols = F.conv2d_backward_data(dy, w1, tuple(cols.shape), 1, 0, 1, 64)
            out["dgrad"] = timed(lambda: F.conv2d_backward_data(dy, w1, tuple(cols.shape), 1, 0, 1, 64))
            args = geom
            out["col2im"] = timed(lambda: L.cpm_deform_col2im(H.ptr(dcols), H.ptr(off), *args, H.ptr(dx), s))
            if deform:
                out["coord"] = timed(lambda: L.cpm_deform_coord_grad(H.ptr(dcols), H.ptr(x), H.ptr(off), *args,
                                                                     H.ptr(doff), s))
            out["wgrad_cols"] = timed(lambda: F.conv2d_backward_weight(cols, dy, w1, 1, 0, 1, 64))
        print("C=%d %dx%d: " % (C, Hh, W) + "  ".join("%s %.0f" % kv for kv in out.items()), flush=True)


if __name__ == "__main__":
    main()
