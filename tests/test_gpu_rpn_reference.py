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


# ---- the head IN TRAINING against the reference: forward, loss and the backward pass the benchmark runs ---------------
@pytest.fixture(scope="module")
def golden_head():
    return np.load(os.path.join(ROOT, "tests", "golden", "rpn_head.npz"))


@pytest.fixture(scope="module")
def meta_head():
    with open(os.path.join(ROOT, "tests", "golden", "rpn_head_meta.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("reductions", ["ordered", "atomic"])
@pytest.mark.parametrize("backward", ["sparse", "dense"])
def test_rpn_head_training_step_equals_reference(golden_head, meta_head, monkeypatch, conv_math, backward, reductions):
    """tests/golden/rpn_head.npz (make_golden.py rpn_head): the REFERENCE's RPNHead.forward -> RPNLossComputation ->
    backward on a 2-image batch over five 256-channel levels, with the reference sampler's own draw (256 per image)
    stored as masks.  Held to it: logits / deltas of every level, both losses, and the gradients of the six head
    parameters and the five feature maps -- through the backward pass over the sampled anchors only
    (csrc/rpn_sparse.hip, the benchmark's default; forced on under ordered reductions too) and through the dense
    formulation, each with ordered slab reductions and with the float-atomic ones the benchmark runs.
    Bars: forward 1e-3 of the tensor maximum (measured 1e-6 f32 / 3e-5 bf16x3), losses 1e-4, every gradient entry
    within 1e-3 of its tensor's maximum (north_star's bar; the atomic mode changes summation order only: same bar)."""
    from detfill import det_fill_, rpn_head_feature
    from pet.lib.ops import _hip, conv as C
    from pet.rcnn.core import config
    from pet.rcnn.modeling.rpn import loss as rpn_loss_mod
    from pet.utils.data.structures.bounding_box import BoxList
    from pet.utils.data.structures.image_list import ImageList
    g, meta = golden_head, meta_head
    config.reset_cfg()
    config.merge_cfg_from_list(CPM_OPTS)
    monkeypatch.setenv("CPM_FUSED_GLUE", "1")
    monkeypatch.setenv("CPM_DEVICE_LISTS", "1")
    prev_sparse = C._RPN_SPARSE
    _hip.set_deterministic(reductions == "ordered")
    C._RPN_SPARSE = 2 if backward == "sparse" else 0      # 2: the sparse pass also under ordered reductions
    try:
        from pet.rcnn.modeling.rpn.rpn import RPNModule
        Hh, W, Cc = meta["H"], meta["W"], meta["C"]
        rpn = RPNModule([Cc] * 5).cuda()
        rpn.train()
        det_fill_(rpn.head)
        if backward == "sparse":
            # (the dense cases keep the module's weights NCHW-contiguous on purpose: every conv call then makes a KRSC
            # copy of its weight, inside the forward side sections too -- the copy must be ordered in front of the side
            # stream's kernel and outlive it: conv._side_copies)
            rpn = rpn.to(memory_format=torch.channels_last)
        shapes = [((Hh + s - 1) // s, (W + s - 1) // s) for s in (4, 8, 16, 32, 64)]
        feats = [rpn_head_feature(i, (2, Cc, h, w)).cuda().contiguous(memory_format=torch.channels_last)
                 .requires_grad_(True) for i, (h, w) in enumerate(shapes)]
        images = ImageList(torch.zeros(2, 3, Hh, W, device="cuda"), [(Hh, W)] * 2)
        targets = []
        for n in range(2):
            t = BoxList(torch.from_numpy(g["gt%d" % n]).cuda(), (W, Hh))
            t.add_field("labels", torch.ones(len(t), dtype=torch.int64, device="cuda"))
            targets.append(t)
        obj, reg = rpn.head(feats, sparse_backward=backward == "sparse")      # (RPNModule.forward's call in training)
        if backward == "sparse":
            assert getattr(obj[0], "_cpm_rpn_sparse", None) is not None, "the head did not run as one node"
        for i in range(5):
            assert rel(obj[i], g["obj%d" % i]) < 1e-3 and rel(reg[i], g["reg%d" % i]) < 1e-3, i
        anchors = rpn.anchor_generator(images, feats)
        # the reference's draw instead of this package's sampler (a different generator): same masks, same quota
        pos = torch.from_numpy(g["pos"]).cuda()
        neg = torch.from_numpy(g["neg"]).cuda()
        quota = torch.from_numpy(g["quota"]).cuda()
        seen = []

        def stored_sample(lab, counts, per_image, frac, *a, **k):
            assert lab.numel() == pos.numel() and per_image == 256
            # the draw must be a legal one for the labels THIS package matched: positives among its positives
            assert bool((lab[pos] >= 1).all()) and bool((lab[neg] == 0).all())
            seen.append(1)
            return pos, neg, quota
        monkeypatch.setattr(rpn_loss_mod.ops, "sample_pos_neg", stored_sample)
        l_obj, l_box = rpn.loss_evaluator(anchors, obj, reg, targets)
        assert seen, "the fused loss path did not ask for a sample"
        assert abs(float(l_obj) - meta["loss_objectness"]) <= 1e-4 * abs(meta["loss_objectness"]), float(l_obj)
        assert abs(float(l_box) - meta["loss_rpn_box_reg"]) <= 1e-4 * abs(meta["loss_rpn_box_reg"]), float(l_box)
        (l_obj + l_box).backward()
        torch.cuda.synchronize()
        worst = 0.0
        for i in range(5):
            got = feats[i].grad
            assert got is not None, i
            s1, s2 = meta["grad_stats"]["feat%d" % i]
            gd = got.double()
            assert abs(float((gd ** 2).sum()) ** 0.5 - s2 ** 0.5) <= 1e-3 * s2 ** 0.5, ("feat", i)
            sub = (got[:, ::8] if i == 0 else got).cpu().numpy()
            want = g["dfeat%d" % i]
            e = float(np.abs(sub - want).max() / (np.abs(want).max() + 1e-30))
            worst = max(worst, e)
            assert e < 1e-3, ("dfeat", i, e)
            # the support: a pixel no sampled anchor's 3x3 window reaches has an exactly zero gradient in the reference
            if backward == "sparse":
                assert np.array_equal(sub != 0, want != 0) or float(np.abs(sub[(want == 0)]).max()) == 0.0, i
        for k, q in rpn.head.named_parameters():
            got = q.grad
            assert got is not None, k
            s1, s2 = meta["grad_stats"][k]
            assert abs(float((got.double() ** 2).sum()) ** 0.5 - s2 ** 0.5) <= 1e-3 * s2 ** 0.5, k
            want = g["dparam::" + k]
            sub = (got[::2, ::2] if k == "conv.weight" else got).cpu().numpy()
            e = float(np.abs(sub - want).max() / (np.abs(want).max() + 1e-30))
            worst = max(worst, e)
            assert e < 1e-3, (k, e)
        log = os.path.join(ROOT, "gpurun_out", "parity_log.txt")
        os.makedirs(os.path.dirname(log), exist_ok=True)
        with open(log, "a") as f:
            f.write("rpn_head_training[%s, %s, %s] worst gradient entry error %.2e of the tensor maximum\n"
                    % (conv_math, backward, reductions, worst))
    finally:
        C._RPN_SPARSE = prev_sparse
        _hip.set_deterministic(False)
        config.reset_cfg()


def rel(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else a
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
