#!/usr/bin/env python3
"""GroupNorm backward error against a float64 reference, next to torch's own fp32 kernel (sanity for the wave kernels)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from pet.lib.ops import conv as C  # noqa: E402

CL = torch.channels_last
for R, hw, g in ((88, 7, 36), (88, 14, 9), (7, 8, 36)):
    torch.manual_seed(0)
    x = torch.randn(R, 576, hw, hw, device="cuda") * 2 + 0.3
    gm = torch.randn(576, device="cuda") * 0.2 + 1
    bt = torch.randn(576, device="cuda") * 0.2
    go = torch.randn(R, 576, hw, hw, device="cuda")
    refs = []
    for dt in (torch.float64, torch.float32):
        xr, gr, br = [t.to(dt).clone().requires_grad_(True) for t in (x, gm, bt)]
        F.relu(F.group_norm(xr, g, gr, br, 1e-5)).backward(go.to(dt))
        refs.append((xr.grad.double(), gr.grad.double(), br.grad.double()))
    xd = x.contiguous(memory_format=CL).requires_grad_(True)
    gd, bd = gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    C.group_norm(xd, gd, bd, g, 1e-5, True).backward(go.contiguous(memory_format=CL))
    ours = (xd.grad.double(), gd.grad.double(), bd.grad.double())

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())
    print("R=%d %dx%d G=%d  ours vs f64: dx %.2e dgamma %.2e dbeta %.2e | torch fp32 vs f64: dx %.2e dgamma %.2e dbeta %.2e" %
          ((R, hw, hw, g) + tuple(rel(o, r) for o, r in zip(ours, refs[0])) + tuple(rel(o, r) for o, r in zip(refs[1], refs[0]))))
