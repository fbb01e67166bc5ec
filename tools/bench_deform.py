#!/usr/bin/env python3
"""Times the deformable-conv gather/scatter kernels on the X-101-64x4d-DCN layer shapes (bs=1, 800x1333)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

from pet.lib.ops import _hip as H  # noqa: E402
import importlib  # noqa: E402
D = importlib.import_module("pet.lib.ops.deform_conv")

CL = torch.channels_last
LAYERS = [("layer2", 512, 100, 168, 1, 64), ("layer2.0_s2", 512, 200, 336, 2, 64), ("layer3", 1024, 50, 84, 1, 64),
          ("layer4", 2048, 25, 42, 1, 64)]


def timeit(f, n=10):
    f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for name, c, h, w, stride, groups in LAYERS:
    x = torch.randn(1, c, h, w, device="cuda").contiguous(memory_format=CL)
    wt = torch.randn(c, c // groups, 3, 3, device="cuda").contiguous(memory_format=CL)
    geom = D._geom(x.shape, wt.shape, stride, 1, 1, groups, 1)
    n, hh, ww, cc, r, s, st, pad, dil, g, dg, p, q = geom
    for kind in ("zero", "rand"):
        off = torch.zeros(1, 18, p, q, device="cuda") if kind == "zero" else torch.randn(1, 18, p, q, device="cuda")
        off = off.contiguous(memory_format=CL)
        cols = D.sample_columns(x, off, geom)
        dcols = torch.randn_like(cols)
        dx = torch.zeros_like(x)
        doff = torch.empty_like(off)
        args = (n, hh, ww, cc, r, s, st, pad, dil, g, dg, p, q)
        t_im = timeit(lambda: D.sample_columns(x, off, geom))
        t_c2 = timeit(lambda: H.lib().cpm_deform_col2im(H.ptr(dcols), H.ptr(off), *args, H.ptr(dx), H.stream()))
        t_cg = timeit(lambda: H.lib().cpm_deform_coord_grad(H.ptr(dcols), H.ptr(x), H.ptr(off), *args, H.ptr(doff),
                                                            H.stream()))
        mb = cols.numel() * 4 / 1e6
        print("%-12s %-5s cols %7.1f MB | im2col %7.1f us | col2im %7.1f us | coord %7.1f us" % (name, kind, mb, t_im,
                                                                                                t_c2, t_cg))
