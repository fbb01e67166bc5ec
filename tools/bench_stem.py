#!/usr/bin/env python3
"""The stem at the BASELINE input size, both forms (bf16x3): python tools/bench_stem.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from bench_conv import timeit  # noqa: E402
from pet.lib.ops import _hip, conv as ops  # noqa: E402

_hip.set_conv_math("bf16x3")
CL = torch.channels_last
x = (torch.randn(2, 3, 800, 1344, device="cuda") * 50).contiguous(memory_format=CL)
w = (torch.randn(64, 3, 7, 7, device="cuda") * 0.05).contiguous(memory_format=CL)
wp = torch.zeros(64, 160, device="cuda")
wp[:, :147] = w.permute(0, 2, 3, 1).reshape(64, 147)
wp = wp.view(64, 160, 1, 1)
sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda")
a = ops.stem_forward(x, wp, sc, sh, w=None)
b = ops.stem_forward(x, wp, sc, sh, w=w)
print("max diff (relative to max):", float((a - b).abs().max() / a.abs().max()))
print("im2col + GEMM + pool: %.1f us" % (timeit(lambda: ops.stem_forward(x, wp, sc, sh, w=None), 20) * 1e3))
print("one kernel   + pool: %.1f us" % (timeit(lambda: ops.stem_forward(x, wp, sc, sh, w=w), 20) * 1e3))
