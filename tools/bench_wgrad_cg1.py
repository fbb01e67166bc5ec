#!/usr/bin/env python3
"""Weight gradient of the grouped ConvTranspose2d 576 -> 9 (k4 s2 p1, groups 9) of Grid_output: wgrad_cg1_kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

from pet.lib.ops import conv as C  # noqa: E402

CL = torch.channels_last
for R in (32, 88, 192):
    x = torch.randn(R, 576, 14, 14, device="cuda").contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(576, 1, 4, 4, device="cuda") * 0.1).contiguous(memory_format=CL).requires_grad_(True)
    y = C.conv_transpose2d(x, w, None, 2, 1, 9, False)
    go = torch.randn_like(y)
    for _ in range(3):
        y.backward(go, retain_graph=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        y.backward(go, retain_graph=True)
    b.record()
    torch.cuda.synchronize()
    print("R=%d: ConvTranspose2d(576->9, g=9) backward (data + weight gradient) %.1f us" % (R, a.elapsed_time(b) / 20 * 1e3))
