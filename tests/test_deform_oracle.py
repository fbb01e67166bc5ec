"""CPU: pin the deformable-conv restatement (oracle/cpm_oracle.c: orc_deform_conv).  The reference op is CUDA-only
(pet/lib/ops/csrc/Deformable/*.cu, no CPU kernel, no vectors), so the oracle is pinned by (1) the known answer
'zero offsets == grouped convolution' against torch's conv2d, values and all three gradients, and (2) an
independent autograd formulation (tests/deform_ref.py) on random offsets that leave the map on every side."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from deform_ref import deform_conv_torch


CASES = [  # N, C, H, W, K, stride, pad, dil, groups, dg
    (2, 8, 7, 9, 8, 1, 1, 1, 2, 1),
    (1, 16, 10, 8, 32, 2, 1, 1, 4, 1),
    (1, 8, 9, 9, 8, 1, 2, 2, 1, 2),
    (2, 12, 6, 11, 6, 2, 1, 1, 3, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_zero_offset_is_grouped_conv(oracle, case):
    N, C, H, W, K, stride, pad, dil, groups, dg = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    w = torch.randn(K, C // groups, 3, 3, generator=g, requires_grad=True)
    y = F.conv2d(x, w, None, stride, pad, dil, groups)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    off = np.zeros((N, dg * 18, y.shape[2], y.shape[3]), np.float32)
    for o in (off, None):
        yo, dx, doff, dw = oracle.deform_conv(x.detach().numpy(), o, w.detach().numpy(), stride, pad, dil, groups, dg,
                                              dy=dy.numpy())
        np.testing.assert_allclose(yo, y.detach().numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(dx, x.grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(dw, w.grad.numpy(), rtol=1e-4, atol=2e-4)
    y0 = oracle.deform_conv(x.detach().numpy(), off, w.detach().numpy(), stride, pad, dil, groups, dg)
    np.testing.assert_array_equal(y0, yo)


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_autograd_formulation(oracle, case):
    N, C, H, W, K, stride, pad, dil, groups, dg = case
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(K, C // groups, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    P = (H + 2 * pad - dil * 2 - 1) // stride + 1
    Q = (W + 2 * pad - dil * 2 - 1) // stride + 1
    # offsets up to +-3 px: many samples cross the border and some leave the (-1, H) x (-1, W) window entirely
    off = (torch.rand(N, dg * 18, P, Q, generator=g, dtype=torch.float64) * 6 - 3).requires_grad_(True)
    y = deform_conv_torch(x, off, w, stride, pad, dil, groups, dg)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    yo, dx, doff, dw = oracle.deform_conv(x.detach().numpy(), off.detach().numpy(), w.detach().numpy(), stride, pad,
                                          dil, groups, dg, dy=dy.numpy())
    np.testing.assert_allclose(yo, y.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dx, x.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dw, w.grad.numpy(), rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(doff, off.grad.numpy(), rtol=1e-3, atol=2e-4)
