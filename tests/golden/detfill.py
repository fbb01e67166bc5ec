"""Deterministic, name-keyed parameter fill shared by the golden generator and the tests.

Every tensor is filled from a numpy RandomState seeded by crc32(name), so two
module trees with the same state-dict keys (the reference's and ours) get
bit-identical weights regardless of construction order.
"""
import zlib

import numpy as np
import torch


def det_tensor(name, shape, kind):
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    shape = tuple(shape)
    if kind == "weight":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        a = rs.standard_normal(shape).astype(np.float32) * np.float32(1.0 / np.sqrt(max(fan_in, 1)))
    elif kind == "scale":      # affine / norm gammas
        a = rs.uniform(0.5, 1.5, shape).astype(np.float32)
    else:                      # biases
        a = rs.uniform(-0.1, 0.1, shape).astype(np.float32)
    return torch.from_numpy(a)


def kind_of(name, tensor):
    if name.endswith("bias"):
        return "bias"
    if tensor.dim() == 1:
        return "scale"
    return "weight"


def det_fill_(module):
    """Through load_state_dict (not in-place writes into state_dict() tensors): a module may keep a tensor in another
    physical layout than its state-dict form (ops.Linear with a window)."""
    fill = {name: det_tensor(name, p.shape, kind_of(name, p)) for name, p in module.state_dict().items()
            if torch.is_floating_point(p)}
    module.load_state_dict(fill, strict=False)
    return module


def soften_heatmaps_(model, factor=0.05):
    """Scale the last transposed conv of every Grid_output so that the heat-map logits stay a few units wide.  With
    the raw name-seeded weights many sigmoids saturate to exactly 1.0 and the per-point arg-max of the grid decoder is
    decided by float ties (first index among equal values) -- a property of one sigmoid implementation's rounding,
    not of the decoder.  Used identically by the golden generator (on the reference model) and by the tests."""
    head = model.Grid_Cascade_RCNN
    s = 0
    with torch.no_grad():
        while hasattr(head, "Output_grid_%d" % s):
            out = getattr(head, "Output_grid_%d" % s)
            out.deconv_2.weight.mul_(factor)
            out.deconv_2.bias.mul_(factor)
            s += 1
    return model


def rpn_head_feature(level, shape):
    """the feature map of FPN level `level` for the rpn_head fixture (make_golden.py rpn_head / test_gpu_rpn_reference):
    numpy's legacy RandomState stream keyed by name, so generator and test build the same bits; signed values of unit
    scale, as FPN outputs are"""
    n, c, h, w = shape
    return det_tensor("rpn_head_fixture.feat%d" % level, shape, "weight") * float(np.sqrt(c * h * w))
