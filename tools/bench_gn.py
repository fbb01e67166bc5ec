#!/usr/bin/env python3
"""GroupNorm(36, 576) + ReLU on [R, 576, 7, 7] (the grid head's 24 norm layers per step): forward and backward time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

from pet.lib.ops import conv as C  # noqa: E402

CL = torch.channels_last


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for R, hw, g in ((32, 7, 36), (88, 7, 36), (192, 7, 36), (88, 14, 9)):
    x = torch.randn(R, 576, hw, hw, device="cuda").contiguous(memory_format=CL).requires_grad_(True)
    gm = torch.ones(576, device="cuda", requires_grad=True)
    bt = torch.zeros(576, device="cuda", requires_grad=True)
    y = C.group_norm(x, gm, bt, g, 1e-5, True)
    go = torch.randn_like(y)
    tf = timeit(lambda: C.group_norm(x, gm, bt, g, 1e-5, True))
    tb = timeit(lambda: y.backward(go, retain_graph=True))
    print("R=%-4d %dx%d GroupNorm(%d, 576): forward %.1f us, backward %.1f us (incl. autograd's accumulation)" % (R, hw, hw, g, tf, tb))
