#!/usr/bin/env python3
"""Every DeformConvPack of the X-50-64x4d + DCN body on the inputs of its exact-f32 run (teacher-forced): the fused
kernels against the column-matrix path, output and the three gradients for one upstream gradient.
    python tools/deform_fused_layers.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cpm-r-cnn_amd", "tests", os.path.join("tests", "golden"), ""):
    sys.path.insert(0, os.path.join(ROOT, p))


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    from test_host_logic import CPM_OPTS
    from test_gpu_deform import X_OPTS
    from detfill import det_fill_
    import pet.lib.ops  # noqa: F401
    from pet.lib.ops import _hip
    dc = sys.modules["pet.lib.ops.deform_conv"]
    from pet.rcnn.core import config
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    config.reset_cfg()
    config.merge_cfg_from_list(CPM_OPTS + X_OPTS)
    _hip.set_conv_math("f32")
    model = convert_bn2affine_model(Generalized_RCNN(is_train=True))
    det_fill_(model)
    model = model.cuda().to(memory_format=torch.channels_last)
    rng = np.random.default_rng(7)
    img = torch.from_numpy(rng.uniform(-100, 150, (1, 3, 96, 128)).astype(np.float32)).cuda()
    img = img.contiguous(memory_format=torch.channels_last)
    layers = [(n, m) for n, m in model.Conv_Body.named_modules() if isinstance(m, dc.DeformConvPack)]
    seen = {}
    hooks = [m.register_forward_hook(lambda mod, args, kwargs, out, n=n: seen.__setitem__(n, (args, kwargs, out)),
                                     with_kwargs=True) for n, m in layers]
    dc.set_fused(False)
    with torch.no_grad():
        model.Conv_Body(img)
    for h in hooks:
        h.remove()
    for n, m in layers:
        args, kwargs, out = seen[n]
        names = ("scale", "shift", "relu")
        scale, shift, relu = [kwargs.get(k, args[1 + i] if len(args) > 1 + i else d)
                              for i, (k, d) in enumerate(zip(names, (None, None, False)))]
        x = args[0].detach()
        with torch.no_grad():
            off = m.conv_offset(x)
        dy = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32)).cuda()
        dy = dy.contiguous(memory_format=torch.channels_last)
        res = []
        for on in (False, True):
            dc.set_fused(on)
            xi, oi = x.clone().requires_grad_(True), off.clone().requires_grad_(True)
            m.weight.grad = None
            y = m._run(xi, oi, scale, shift, relu, False)
            y.backward(dy)
            res.append((y.detach(), xi.grad, oi.grad, m.weight.grad.clone()))
        a, b = res
        valid = float(((off.abs() < 2).float().mean()))
        print("%-16s x %s stride %d |off|<2: %.2f  y %.2e  dx %.2e  doff %.2e  dw %.2e" % (
            n, tuple(x.shape), m.stride[0], valid, rel(b[0], a[0]), rel(b[1], a[1]), rel(b[2], a[2]), rel(b[3], a[3])))


if __name__ == "__main__":
    main()
