"""GPU: the offset-regression Cascade R-CNN head with ISM + RSM against the REFERENCE modules under
identical name-keyed deterministic weights (tests/golden/model_cascade.npz, written by make_golden.py cascade):
per-stage logits, the evaluation path (decode / refine / ensemble / IoU-merged scores), and the training path on a
proposal set the sampler keeps whole (losses and gradient statistics)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from test_host_logic import CASCADE_OPTS

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def rel(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else a
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "model_cascade.npz"))


@pytest.fixture(scope="module")
def model():
    from detfill import det_fill_
    from pet.rcnn.core import config
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    config.reset_cfg()
    config.merge_cfg_from_list(CASCADE_OPTS)
    m = convert_bn2affine_model(Generalized_RCNN(is_train=True))
    det_fill_(m)
    m = m.cuda().to(memory_format=CL)
    yield m
    config.reset_cfg()


def _props(g):
    from pet.utils.data.structures.bounding_box import BoxList
    b = BoxList(torch.from_numpy(np.concatenate([g["rois"], g["gt"]])).cuda(), (96, 64))
    b.add_field("objectness", torch.linspace(0.9, 0.1, len(b)).cuda())
    return [b]


def _targets(g):
    from pet.utils.data.structures.bounding_box import BoxList
    t = BoxList(torch.from_numpy(g["gt"]).cuda(), (96, 64))
    t.add_field("labels", torch.from_numpy(g["gt_labels"]).cuda())
    return [t]


def test_cascade_eval_matches_reference(model, golden, conv_math):
    g = golden
    model.eval()
    head = model.Cascade_RCNN
    with torch.no_grad():
        p = model.Conv_Body_FPN(model.Conv_Body(torch.from_numpy(g["img"]).cuda()))
        for s in (1, 2):
            f = getattr(head, "Box_Head_%d" % s)(p, _props(g))
            c, b, i = getattr(head, "Output_%d" % s)(f)
            assert rel(c, g["s%d_cls" % s]) < 1e-3 and rel(b, g["s%d_bbox" % s]) < 1e-3
            if s == 2:
                assert rel(i, g["s2_iou"]) < 1e-3
            else:
                assert i is None
        x, result, losses = head(p, _props(g))
        assert losses == {} and len(result) == 1
        assert rel(x, g["eval_x"]) < 1e-3
        assert result[0].bbox.shape == g["eval_bbox"].shape == (15 * 81, 4)
        # decoded boxes: image-scale coordinates (clipped to 96 x 64); scores: softmax of the 2-stage ensemble x IoU
        assert float(np.abs(result[0].bbox.cpu().numpy() - g["eval_bbox"]).max()) < 0.05
        # the name-seeded weights give class logits and box deltas tens of units wide: the second stage pools from boxes
        # decoded through exp() of those deltas and the softmax amplifies the logits' relative error (f32 ~3e-6,
        # split-bf16 ~3e-5 of the largest logit) by the logits' width; run-to-run (float atomics) 2e-3 .. 7e-3
        tol = 2e-3 if conv_math == "f32" else 1e-2
        assert float(np.abs(result[0].get_field("scores").cpu().numpy() - g["eval_scores"]).max()) < tol


def test_cascade_training_matches_reference(model, golden, conv_math):
    g = golden
    with open(os.path.join(ROOT, "tests", "golden", "model_cascade_meta.json")) as f:
        meta = json.load(f)
    model.train()
    for q in model.parameters():
        q.grad = None
    head = model.Cascade_RCNN
    p = model.Conv_Body_FPN(model.Conv_Body(torch.from_numpy(g["img"]).cuda()))
    x, proposals, losses = head(p, _props(g), _targets(g))
    want = {k[6:]: float(g[k]) for k in g.files if k.startswith("loss::")}
    assert set(losses) == set(want)
    for k, v in want.items():
        got = float(losses[k].detach())
        assert abs(got - v) <= 2e-3 * abs(v) + 1e-4, (k, got, v)
    assert np.array_equal(proposals[0].get_field("labels").cpu().numpy(), g["train_final_labels"])
    assert float(np.abs(proposals[0].bbox.detach().cpu().numpy() - g["train_final_bbox"]).max()) < 0.05
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    params = dict(model.named_parameters())
    checked = 0
    for k, (s1, sabs, s2) in meta["grad_stats"].items():
        gr = params[k].grad
        assert gr is not None, k
        gr = gr.double()
        assert abs(float((gr ** 2).sum()) ** 0.5 - s2 ** 0.5) <= 5e-3 * s2 ** 0.5 + 1e-6, k
        assert abs(float(gr.abs().sum()) - sabs) <= 5e-3 * sabs + 1e-6, k
        checked += 1
    assert checked >= 20


def test_cascade_end_to_end_training_steps():
    """RPN -> 2-stage cascade -> ISM -> RSM from images, with the flat-buffer SGD: losses finite, every trainable
    tensor receives a gradient, parameters move."""
    from pet.rcnn.core import config
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    from pet.utils.optimizer import Optimizer
    from test_gpu_model import synthetic_batch
    config.reset_cfg()
    config.merge_cfg_from_list(CASCADE_OPTS)
    try:
        torch.manual_seed(0)
        m = convert_bn2affine_model(Generalized_RCNN(is_train=True)).cuda().to(memory_format=CL)
        m.train()
        opt = Optimizer(m, config.cfg.SOLVER).build()
        for g_ in opt.param_groups:
            g_["lr"] = 1e-3 * g_["lr_scale"]
        images, targets = synthetic_batch(2, 256, 320, 6, seed=3)
        images, targets = images.cuda().contiguous(memory_format=CL), [t.to("cuda") for t in targets]
        before = m.Cascade_RCNN.Output_2.cls_score.weight.detach().clone()
        for it in range(2):
            opt.zero_grad()
            losses = m(images, targets)["losses"]
            assert set(losses) == {"loss_objectness", "loss_rpn_box_reg", "s1_cls_loss", "s1_bbox_loss", "s2_cls_loss",
                                   "s2_bbox_loss", "loss_iou_2", "loss_rescore"}
            total = sum(losses.values())
            assert torch.isfinite(total)
            total.backward()
            if it == 0:
                for k, q in m.named_parameters():
                    if q.requires_grad:
                        assert q.grad is not None and torch.isfinite(q.grad).all(), k
            opt.step()
        assert not torch.equal(before, m.Cascade_RCNN.Output_2.cls_score.weight.detach())
    finally:
        config.reset_cfg()
