#!/usr/bin/env python3
"""Data gradient of a 3x3 stride-1 conv with and without the fused ReLU-gate / scale epilogue (halo kernel)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

from pet.lib.ops import conv as C  # noqa: E402


def timeit(fn, n=30):
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


CL = torch.channels_last
for (N, H, W, Cc, K, R) in ((2, 100, 168, 128, 128, 3), (2, 50, 84, 256, 256, 3), (2, 25, 42, 512, 512, 3),
                           (2, 100, 168, 128, 512, 1), (2, 50, 84, 256, 1024, 1), (2, 25, 42, 512, 2048, 1)):
    dev = "cuda"
    w = (torch.randn(K, Cc, R, R, device=dev) * 0.05).contiguous(memory_format=CL)
    dy = torch.randn(N, K, H, W, device=dev).contiguous(memory_format=CL)
    act = torch.randn(N, Cc, H, W, device=dev).contiguous(memory_format=CL)
    scale = torch.rand(Cc, device=dev) + 0.5
    t0 = timeit(lambda: C.conv2d_backward_data(dy, w, (N, Cc, H, W), 1, R // 2, 1, 1))
    t1 = timeit(lambda: C.conv2d_backward_data_gated(dy, w, act, scale, 1, R // 2, 1, 1))
    gf = 2.0 * N * H * W * K * R * R * Cc / 1e9
    print("%-28s plain %6.1f us (%5.1f TF)  gate+scale %6.1f us (%5.1f TF)" % ((N, H, W, Cc, K, R), t0, gf / t0 * 1e3, t1,
                                                                            gf / t1 * 1e3))
