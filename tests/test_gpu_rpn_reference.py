"""GPU: row a-3 as wholes against the REFERENCE itself (tests/golden/rpn_whole.npz, written by `make_golden.py rpn`).

The fixture holds the reference's RPNLossComputation.__call__ (rpn/loss.py:88-126) and RPNPostProcessor.forward in
training mode (rpn/inference.py:67-172) on a 2-image batch over five FPN levels: both losses, their gradients w.r.t.
every level's objectness / delta map, and the proposals (boxes + objectness, gt boxes appended).  The sampler budget
(cfg.RPN.BATCH_SIZE_PER_IMAGE) is raised until every valid anchor is kept, so no random draw is involved; NMS went
through the oracle's greedy NMS, which tests/test_oracle_golden.py pins to the reference's compiled soft_nms.cpp.

Held to it: the fused loss (cpm_match_rois + cpm_rpn_labels + the batch sampler + cpm_rpn_loss: one launch each), the
per-image tensor-op loss, and the three proposal paths -- packed device list (cpm_proposals_finalize), batch-fused with
host index lists, per-level / per-image."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from test_host_logic import CPM_OPTS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "rpn_whole.npz"))


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(ROOT, "tests", "golden", "rpn_whole_meta.json")) as f:
        return json.load(f)


@pytest.fixture()
def rpn_cfg(meta):
    from pet.rcnn.core import config
    config.reset_cfg()
    config.merge_cfg_from_list(CPM_OPTS)
    config.merge_cfg_from_list(["RPN.BATCH_SIZE_PER_IMAGE", 1000000, "RPN.PRE_NMS_TOP_N_TRAIN", meta["pre_nms"],
                                "RPN.POST_NMS_TOP_N_TRAIN", meta["post_nms"],
                                "RPN.FPN_POST_NMS_TOP_N_TRAIN", meta["fpn_post_nms"]])
    yield config.cfg
    config.reset_cfg()


def _build(monkeypatch, fused, lists):
    monkeypatch.setenv("CPM_FUSED_GLUE", "1" if fused else "0")
    monkeypatch.setenv("CPM_DEVICE_LISTS", "1" if lists else "0")
    from pet.rcnn.modeling.rpn.rpn import RPNModule
    rpn = RPNModule([256] * 5).cuda()
    rpn.train()
    return rpn


def _inputs(g, meta, rpn):
    from pet.utils.data.structures.bounding_box import BoxList
    from pet.utils.data.structures.image_list import ImageList
    H, W = meta["H"], meta["W"]
    obj = [torch.from_numpy(g["obj%d" % i]).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
           for i in range(5)]
    reg = [torch.from_numpy(g["reg%d" % i]).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
           for i in range(5)]
    feats = [torch.zeros(2, 4, o.shape[2], o.shape[3], device="cuda") for o in obj]
    images = ImageList(torch.zeros(2, 3, H, W, device="cuda"), [(H, W)] * 2)
    targets = []
    for n in range(2):
        t = BoxList(torch.from_numpy(g["gt%d" % n]).cuda(), (W, H))
        t.add_field("labels", torch.ones(len(t), dtype=torch.int64, device="cuda"))
        targets.append(t)
    anchors = rpn.anchor_generator(images, feats)
    return anchors, obj, reg, targets


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per_image"])
def test_rpn_loss_equals_reference(golden, meta, rpn_cfg, monkeypatch, fused):
    """Both losses to 1e-5 relative and the gradient of their sum w.r.t. every objectness / delta map to 1e-6 absolute
    (the entries are O(1e-4): 1 / sample size) -- values AND support: an anchor the reference did not sample has an
    exactly zero gradient here too."""
    rpn = _build(monkeypatch, fused, fused)
    anchors, obj, reg, targets = _inputs(golden, meta, rpn)
    l_obj, l_box = rpn.loss_evaluator(anchors, obj, reg, targets)
    assert abs(float(l_obj) - meta["loss_objectness"]) <= 1e-5 * abs(meta["loss_objectness"]), float(l_obj)
    assert abs(float(l_box) - meta["loss_rpn_box_reg"]) <= 1e-5 * abs(meta["loss_rpn_box_reg"]) + 1e-9, float(l_box)
    (l_obj + l_box).backward()
    for i in range(5):
        for name, t in (("dobj", obj[i]), ("dreg", reg[i])):
            want = golden["%s%d" % (name, i)]
            got = t.grad.cpu().numpy()
            assert np.abs(got - want).max() <= 1e-6, (name, i, float(np.abs(got - want).max()))
            assert np.array_equal(got != 0, want != 0), (name, i, "support of the gradient (the sampled anchors)")


def _boxlists(props, n_img):
    """per-image (boxes, objectness) numpy arrays from a list[BoxList] or a packed device RoIList"""
    if not isinstance(props, (list, tuple)):
        props = props.to_boxlists()
    assert len(props) == n_img
    return [(p.bbox.cpu().numpy(), p.get_field("objectness").cpu().numpy()) for p in props]


@pytest.mark.parametrize("path", ["device_list", "fused_host_lists", "per_level"])
def test_rpn_proposals_equal_reference(golden, meta, rpn_cfg, monkeypatch, path):
    """Same proposals in the same order: objectness bit-equal (sigmoid of the same logits; it decides every top-k and
    the NMS order), boxes to 1e-3 px (expf of the device vs the host's exp in BoxCoder.decode)."""
    rpn = _build(monkeypatch, path != "per_level", path == "device_list")
    anchors, obj, reg, targets = _inputs(golden, meta, rpn)
    sel = rpn.box_selector_train
    with torch.no_grad():
        if path == "device_list":
            assert sel.can_keep_on_device(obj, targets)
            props = sel.finish_device(sel.start_fused(anchors, obj, reg, read_counts=False), targets)
        else:
            props = sel(anchors, obj, reg, targets)
    got = _boxlists(props, 2)
    for n in range(2):
        wb, wo = golden["prop_box%d" % n], golden["prop_obj%d" % n]
        gb, go = got[n]
        assert gb.shape == wb.shape, (n, gb.shape, wb.shape)
        assert np.array_equal(go, wo), (n, "objectness / order", float(np.abs(go - wo).max()))
        assert np.abs(gb - wb).max() <= 1e-3, (n, float(np.abs(gb - wb).max()))
