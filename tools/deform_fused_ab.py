#!/usr/bin/env python3
"""X-50-64x4d + DCN body (tests/test_gpu_deform.py's fixture): features and every trainable gradient with the fused
deformable kernels against the column-matrix path on the same weights and image, exact-f32 arithmetic.
    python tools/deform_fused_ab.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)


def main():
    from test_host_logic import CPM_OPTS
    from test_gpu_deform import X_OPTS
    from detfill import det_fill_
    import pet.lib.ops  # noqa: F401
    from pet.lib.ops import _hip
    deform_conv = sys.modules["pet.lib.ops.deform_conv"]
    from pet.rcnn.core import config
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    config.reset_cfg()
    config.merge_cfg_from_list(CPM_OPTS + X_OPTS)
    _hip.set_conv_math("f32")
    model = convert_bn2affine_model(Generalized_RCNN(is_train=True))
    det_fill_(model)
    model = model.cuda().to(memory_format=torch.channels_last).train()
    rng = np.random.default_rng(7)
    img = torch.from_numpy(rng.uniform(-100, 150, (1, 3, 96, 128)).astype(np.float32)).cuda()
    img = img.contiguous(memory_format=torch.channels_last)
    out = {}
    for on in (False, True, True):
        deform_conv.set_fused(on)
        model.zero_grad(set_to_none=True)
        feats = model.Conv_Body_FPN(model.Conv_Body(img))
        sum(f.square().mean() for f in feats).backward()
        key = "fused" if on else "cols"
        if key in out:
            key = "fused2"
        out[key] = ([f.detach().clone() for f in feats],
                    {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    for other in ("fused", "fused2"):
        base = "cols" if other == "fused" else "fused"
        print("== %s vs %s" % (other, base))
        for i, (a, b) in enumerate(zip(out[other][0], out[base][0])):
            print("feature %d: max rel %.3g" % (i, float((a - b).abs().max() / b.abs().max())))
        worst = []
        for k, g in out[base][1].items():
            h = out[other][1][k]
            worst.append((float((g - h).norm() / (g.norm() + 1e-30)), float(g.norm()), float(h.norm()), k))
        worst.sort(reverse=True)
        for w in worst[:12]:
            print("  L2 rel %.3g  norms %.6g %.6g  %s" % w)


if __name__ == "__main__":
    main()
