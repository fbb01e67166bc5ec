"""GPU: rows a-11 / a-12 against the REFERENCE itself (tests/golden/model_cpm.npz, written by `make_golden.py cpm`).

The fixture holds a training-mode forward of the reference's GridCascadeRCNN (grid_cascade_rcnn.py:57-224) on a
2-image batch whose proposal sets every sampler keeps whole (no random draw): the RoI set entering each grid stage,
the cls / RSM samples, the 8 losses and gradient statistics -- and CLSPostProcessor's candidate selection
(inference.py:59-124).  The packed device-list path the training step runs on (csrc/roi_lists.hip), the batch-fused
path with host index lists (cpm_match_rois / cpm_grid_bce_loss / cpm_grid_decode) and the per-image formulation are ALL
held to it, in both conv arithmetics."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from test_host_logic import CPM_OPTS

pytestmark = pytest.mark.gpu
CL = torch.channels_last
W, H = 224, 160


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "model_cpm.npz"))


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(ROOT, "tests", "golden", "model_cpm_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def model():
    from detfill import det_fill_, soften_heatmaps_
    from pet.rcnn.core import config
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    config.reset_cfg()
    config.merge_cfg_from_list(CPM_OPTS)
    m = convert_bn2affine_model(Generalized_RCNN(is_train=True))
    det_fill_(m)
    soften_heatmaps_(m)
    m = m.cuda().to(memory_format=CL)
    yield m
    config.reset_cfg()


def _inputs(g):
    from pet.utils.data.structures.bounding_box import BoxList
    props, targets = [], []
    for i in range(2):
        b = BoxList(torch.from_numpy(g["props_%d" % i]).cuda(), (W, H))
        b.add_field("objectness", torch.linspace(0.95, 0.05, len(b)).cuda())
        props.append(b)
        t = BoxList(torch.from_numpy(g["gt_%d" % i]).cuda(), (W, H))
        t.add_field("labels", torch.from_numpy(g["gt_labels_%d" % i]).cuda())
        targets.append(t)
    return props, targets


def _same_boxes(got, want, what, tol=0.05, flips_allowed=0):
    """Boxes equal within `tol` pixels.  Decoded boxes (stages >= 1) come from a per-point arg-max over a smooth
    28x28 heat map whose best and second-best cells can be closer (fixture meta: min_argmax_gap, in units of the
    largest logit) than the split-bf16 arithmetic's ~3e-5 forward error; a flipped arg-max moves one of a side's
    three voting points by one cell, i.e. the box side by at most ~ (1 + ratio) * extent / 56.  Such RoIs are COUNTED
    (`flips_allowed`, 0 in exact-f32 arithmetic) and bounded to that size; everything else must agree to `tol`."""
    got = got.detach().cpu().numpy()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not got.size:
        return 0
    dev = np.abs(got - want).max(axis=1)
    extent = np.maximum(want[:, 2] - want[:, 0], want[:, 3] - want[:, 1])
    flipped = dev >= tol
    assert int(flipped.sum()) <= flips_allowed, (what, "RoIs off by more than %g px: %d" % (tol, int(flipped.sum())),
                                                  float(dev.max()))
    assert bool((dev[flipped] <= 0.05 * extent[flipped] + tol).all()), (what, float(dev.max()))
    return int(flipped.sum())


@pytest.mark.parametrize("fused", ["lists", True, False], ids=["device_lists", "fused_glue", "per_image"])
def test_cpm_train_forward_matches_reference(model, golden, meta, conv_math, fused, reductions):
    g = golden
    head = model.Grid_Cascade_RCNN
    saved = (head.fused_glue, head.cls_loss_evaluator.fused_glue, head.rescore_loss_evaluator.fused_glue)
    head.fused_glue = head.cls_loss_evaluator.fused_glue = head.rescore_loss_evaluator.fused_glue = bool(fused)
    stage_rois = []
    heads = [getattr(head, "Head_grid_%d" % s) for s in range(3)]
    hooks = [h.register_forward_pre_hook(lambda m, a: stage_rois.append([b.bbox.detach().clone() for b in a[1]]))
             for h in heads]
    try:
        model.train()
        for q in model.parameters():
            q.grad = None
        props, targets = _inputs(g)
        if fused == "lists":                     # the packed device lists the training step runs on (roi_lists.hip)
            from pet.lib.ops import roi_lists as RL
            assert head.takes_device_lists
            props = RL.from_boxlists(props)
        feats = model.Conv_Body_FPN(model.Conv_Body(torch.from_numpy(g["img"]).cuda()))
        x, result, losses = head(feats, props, targets)
        # --- the samples: cls head, every grid stage, RSM head ------------------------------------------------
        cls_props = head.cls_loss_evaluator._proposals
        for i in range(2):
            _same_boxes(cls_props[i].bbox, g["cls_rois_%d" % i], "cls sample img %d" % i, tol=1e-6)
            assert np.array_equal(cls_props[i].get_field("labels").cpu().numpy(), g["cls_labels_%d" % i])
        assert len(stage_rois) == 3
        budget = 0 if conv_math == "f32" else 2
        flips = 0
        for s in range(3):
            for i in range(2):
                # stage 0 RoIs are input proposals (exact); later stages are decoded from heat maps (pixel scale)
                flips += _same_boxes(stage_rois[s][i], g["stage%d_rois_%d" % (s, i)], "stage %d img %d" % (s, i),
                                     tol=1e-6 if s == 0 else 0.05, flips_allowed=0 if s == 0 else budget)
        for i in range(2):
            _same_boxes(result[i].bbox, g["rescore_rois_%d" % i], "rescore sample img %d" % i, flips_allowed=budget)
            assert np.array_equal(result[i].get_field("labels").cpu().numpy(), g["rescore_labels_%d" % i])
        # --- the 8 losses (6 here: RPN is not part of the head) -----------------------------------------------
        want = {k[6:]: float(g[k]) for k in g.files if k.startswith("loss::")}
        assert set(losses) == set(want)
        loss_tol = 1e-3 if flips == 0 else 1e-2          # a moved RoI moves its own target and heat-map term
        for k, v in want.items():
            got = float(losses[k].detach())
            assert abs(got - v) <= loss_tol * abs(v), (k, got, v)
        assert float(np.abs(x.detach().cpu().numpy()[:, ::16] - g["last_x"]).max()) <= 1e-3 * float(np.abs(g["last_x"]).max())
        # --- gradients: L1 / L2 norms of every tensor the head's losses reach ---------------------------------
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        params = dict(model.named_parameters())
        checked, worst = 0, 0.0
        for k, (s1, sabs, s2) in meta["grad_stats"].items():
            gr = params[k].grad
            assert gr is not None, k
            gr = gr.double()
            e2 = abs(float((gr ** 2).sum()) ** 0.5 - s2 ** 0.5) / (s2 ** 0.5 + 1e-30)
            e1 = abs(float(gr.abs().sum()) - sabs) / (sabs + 1e-30)
            worst = max(worst, e1, e2)
            # split-bf16 products differ from the reference's fp32 ones in the last bits: a pre-activation within
            # that distance of 0 gates differently, which moves ONE entry of a gradient whose norm is carried by a
            # few dozen RoIs.  With float atomics those bits also varied from run to run (up to 2.4e-3 of a norm, a
            # different tensor each time): the test now runs with ordered reductions, which removes that noise.
            # deterministic reductions (fixture): measured 1.9e-4 (f32) / 5.1e-4 (bf16x3), no flips, identical on
            # all three paths and from run to run -- bounds at 1e-3 / 2e-3; a counted arg-max flip keeps the wide one
            # float-atomic reductions (`reductions` == "atomic", the benchmark's default): up to 2.4e-3 of a norm was
            # seen run to run in round 2 -- stated bound 5e-3 in the split-bf16 arithmetic, 2e-3 in exact f32
            if reductions == "ordered":
                tol = 2e-2 if flips else (1e-3 if conv_math == "f32" else 2e-3)
            else:
                tol = 2e-2 if flips else (2e-3 if conv_math == "f32" else 5e-3)
            assert e1 < tol and e2 < tol, (k, e1, e2)
            checked += 1
        assert checked >= 150
        from test_gpu_model import _log
        _log("cpm_reference[%s, %s, %s] flips %d, worst gradient-norm err %.2e" % (conv_math, fused, reductions, flips, worst))
    finally:
        for h in hooks:
            h.remove()
        head.fused_glue, head.cls_loss_evaluator.fused_glue, head.rescore_loss_evaluator.fused_glue = saved


def test_cls_post_processor_candidates_match_reference(golden, meta, monkeypatch):
    """CLSPostProcessor (inference.py:59-124): softmax, boxes repeated per class, clip, score > 0.03 & label != 0 --
    the candidate set handed to ml_nms equals the reference's, then the kept set equals the oracle's ml_nms (pinned to
    the reference's ml_soft_nms.cpp) on the REFERENCE candidates; plus the RSM re-scoring branch."""
    from oracle import pyoracle as O
    from pet.rcnn.core import config
    from pet.utils.data.structures.bounding_box import BoxList
    import pet.rcnn.modeling.grid_cascade_rcnn.inference as inf
    g = golden
    config.reset_cfg()
    config.merge_cfg_from_list(CPM_OPTS)
    try:
        post = inf.post_processor(type="cls")
        assert abs(post.score_thresh - meta["post_thresh"]) < 1e-12 and abs(post.nms - meta["post_nms"]) < 1e-12
        rec = {}
        real_nms = inf.boxlist_ml_nms

        def recording(boxlist, thresh, *a, **k):
            rec["bbox"], rec["scores"] = boxlist.bbox.cpu().numpy(), boxlist.get_field("scores").cpu().numpy()
            rec["labels"] = boxlist.get_field("labels").cpu().numpy()
            return real_nms(boxlist, thresh, *a, **k)
        monkeypatch.setattr(inf, "boxlist_ml_nms", recording)
        logits = torch.from_numpy(g["post_logits"]).cuda()
        res = post(logits, [BoxList(torch.from_numpy(g["post_boxes"]).cuda(), (1333, 800))])
        assert np.array_equal(rec["labels"], g["post_cand_labels"])          # the candidate MASK is identical
        assert np.array_equal(rec["bbox"], g["post_cand_bbox"])              # clipped boxes: pure copies / clamps
        assert float(np.abs(rec["scores"] - g["post_cand_scores"]).max()) < 1e-6
        keep = O.ml_nms(g["post_cand_bbox"], g["post_cand_scores"], g["post_cand_labels"], post.nms)
        assert np.array_equal(res[0].bbox.cpu().numpy(), g["post_cand_bbox"][keep])
        assert np.array_equal(res[0].get_field("labels").cpu().numpy(), g["post_cand_labels"][keep])
        # RSM re-scoring
        bl = BoxList(torch.from_numpy(g["post_boxes"]).cuda(), (1333, 800))
        bl.add_field("scores", torch.from_numpy(g["post_rs_scores_in"]).cuda())
        bl.add_field("labels", torch.from_numpy(g["post_rs_labels"]).cuda())
        r2 = post(logits, [bl], rescore=True)
        got = r2[0].get_field("scores").cpu().numpy()
        assert float(np.abs(got - g["post_rs_scores_out"]).max()) < 1e-6
    finally:
        config.reset_cfg()
