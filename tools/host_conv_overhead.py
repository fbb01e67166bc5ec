#!/usr/bin/env python3
"""Host-side cost of one conv / GroupNorm op call (forward, and forward+backward through autograd) on tiny tensors,
where the device work is negligible: what the launch-bound RoI-head phase of the step pays per op."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpm-r-cnn_amd"))
import torch  # noqa: E402

import pet.lib.ops as ops  # noqa: E402
from pet.lib.ops import _hip  # noqa: E402

_hip.set_conv_math("bf16x3")
conv = ops.Conv2d(64, 64, 3, 1, 1).cuda().to(memory_format=torch.channels_last)
gn = ops.GroupNorm(4, 64).cuda()
x = torch.randn(4, 64, 7, 7, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)


def fwd():
    return gn(conv(x), relu=True)


def fwd_bwd():
    fwd().sum().backward()


for name, fn, n in (("conv+GN forward", fwd, 2000), ("conv+GN forward+backward", fwd_bwd, 1000)):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    print("%-28s %.1f us per call (host)" % (name, dt / n * 1e6))
with torch.no_grad():
    t = time.perf_counter()
    for _ in range(2000):
        fwd()
    print("%-28s %.1f us per call (host)" % ("conv+GN forward, no_grad", (time.perf_counter() - t) / 2000 * 1e6))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    fwd()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
