"""CPU-only: pins oracle/cpu_model.py (the CPU baseline / whole-network checker) to outputs of the reference
model under the same name-keyed deterministic weights (tests/golden/model_r50.npz)."""
import json
import os

import numpy as np
import torch

from conftest import ROOT


def det_state_dict():
    from detfill import det_tensor
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "model_r50_meta.json")))
    sd = {}
    for k, shape in meta["state_dict"]:
        if "cell_anchors" in k:
            continue
        kind = "bias" if k.endswith("bias") else ("scale" if len(shape) == 1 else "weight")
        sd[k] = det_tensor(k, shape, kind)
    return sd


def rel(a, b):
    a = a.detach().numpy()
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_cpu_model_matches_reference(golden_model):
    from oracle import cpu_model as M
    g = golden_model
    sd = det_state_dict()
    with torch.no_grad():
        c = M.backbone(sd, torch.from_numpy(g["m_img"]))
        for i, t in enumerate(c):
            assert rel(t[:, ::8], g["m_c%d" % (i + 2)]) < 1e-5
        p = M.fpn(sd, c)
        for i, t in enumerate(p):
            assert rel(t[:, ::8], g["m_p%d" % (i + 2)]) < 1e-5
        lo, br = M.rpn_head(sd, p)
        for i in range(5):
            assert rel(lo[i], g["m_rpn_logits_%d" % i]) < 1e-5 and rel(br[i], g["m_rpn_bbox_%d" % i]) < 1e-5
        rois = torch.cat([torch.zeros(6, 1), torch.from_numpy(g["m_rois"])], 1)
        assert rel(M.cls_head(sd, p, rois), g["m_cls_logits"]) < 1e-5
        assert rel(M.cls_head(sd, p, rois, "Head_rescore", "Output_rescore"), g["m_rescore_logits"]) < 1e-5
        for s in range(3):
            x, heat, iou = M.grid_stage(sd, p, rois, s, last=(s == 2))
            assert rel(x[:, ::16], g["m_grid_feat_%d" % s]) < 1e-5
            assert rel(heat, g["m_grid_heat_%d" % s]) < 1e-5
            if s == 2:
                assert rel(iou, g["m_grid_iou_2"]) < 1e-5
