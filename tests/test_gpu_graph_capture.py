"""GPU: the static part of the training step as hipGraphs (Generalized_RCNN.capture_static_part) next to the flat
optimizer's "gradients cleared by the SGD kernel" mode: the warm-up iterations of a capture run
real backward passes between step() and the next zero_grad(), so that zero_grad must clear for real."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_zero_grad_skips_its_memset_only_when_nothing_was_written_since_the_step():
    from pet.lib.ops import conv as C
    from pet.utils.optimizer import FlatSGD
    ps = [torch.nn.Parameter(torch.randn(64, 32, device="cuda")), torch.nn.Parameter(torch.randn(64, device="cuda"))]
    opt = FlatSGD([("w", ps[0], 0), ("b", ps[1], 1)], [dict(weight_decay=0.0, lr_scale=1), dict(weight_decay=0.0, lr_scale=2),
                                                       dict(weight_decay=0.0, lr_scale=1)], 0.9)
    opt.clear_grads_in_step = True
    for g in opt.param_groups:
        g["lr"] = 0.1
    opt.flat_grad.fill_(1.0)
    opt.step()                                       # the kernel clears every gradient element behind its use
    torch.cuda.synchronize()
    for b, e in zip(opt.seg_begin.tolist(), opt.seg_end.tolist()):
        assert not opt.flat_grad[b:e].any()
    # nothing ran since: zero_grad trusts the kernel and skips its memset
    opt.flat_grad[0] = 3.0                           # a write the optimizer cannot know about ...
    opt.zero_grad()
    assert float(opt.flat_grad[0]) == 3.0            # ... survives the skipped memset (that is the fast path)
    # a backward route announced itself (every in-place sink counts a use in its forward): the memset runs
    opt.step()
    opt.flat_grad[0] = 3.0
    C._note_use(ps[0])
    opt.zero_grad()
    assert not opt.flat_grad.any()
    assert ps[0]._cpm_uses == 0


def test_gradients_left_by_a_capture_are_cleared_by_the_next_zero_grad():
    import __graft_entry__ as entry
    entry.ensure_built()
    from bench import Trainer, calibrate_frozen_affine, synthetic_batch
    from pet.lib.ops import _hip
    device = torch.device("cuda", 0)
    prev = _hip.get_conv_math()
    _hip.set_conv_math("bf16x3")
    try:
        tr = Trainer(device)
        tr.optimizer.clear_grads_in_step = True
        images, targets = synthetic_batch(2, 192, 256, 5, 11, device)
        calibrate_frozen_affine(tr.model, images.tensors)
        tr.step(images, targets)                     # the data-gradient weight images exist from the first step on
        tr.model.capture_static_part(images.tensors)
        torch.cuda.synchronize()
        opt = tr.optimizer
        assert float(opt.flat_grad.abs().max()) > 0, "warm-up and capture leave gradients in the parameters' sinks"
        opt.zero_grad()
        torch.cuda.synchronize()
        assert not opt.flat_grad.any(), "zero_grad after a capture must clear what the capture's backward passes left"
        for _ in range(2):                           # replayed steps train on
            tr.step(images, targets)
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(v.detach()).all()) for v in tr.last_losses.values())
        assert bool(torch.isfinite(opt.flat_param).all())
    finally:
        _hip.set_conv_math(prev)


def _dets(res):
    r = res[0]
    return r.bbox.clone(), r.get_field("scores").clone(), r.get_field("labels").clone()


def _same(a, b):
    """bit-equal, NaN == NaN (the reference's s ** 0.8 of a negative ISM-merged score is NaN, and so is ours)"""
    if a.shape != b.shape:
        return False
    if a.dtype.is_floating_point:
        a, b = torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)
    return bool(torch.equal(a, b))


def test_test_time_forward_replays_the_static_part_as_a_graph_with_identical_detections(monkeypatch):
    """Generalized_RCNN._eval_static: in eval mode under no_grad the backbone, FPN and RPN head run as one hipGraph per
    input shape.  Held: the detections of the replayed forward equal the eager ones bit for bit, on a second image too
    (the input is copied into the captured buffer), and a parameter changed in place (its version counter moves, as
    load_state_dict does) is seen by the next forward (the graph is re-captured: a replay of the old one would read the
    stale bf16x3 weight image)."""
    import __graft_entry__ as entry
    entry.ensure_built()
    from bench import Trainer, calibrate_frozen_affine, inference_leg, synthetic_batch
    from pet.lib.ops import _hip
    from pet.rcnn.core import config
    device = torch.device("cuda", 0)
    prev = _hip.get_conv_math()
    _hip.set_conv_math("bf16x3")
    _hip.set_deterministic(True)
    try:
        tr = Trainer(device)
        images, _ = synthetic_batch(2, 320, 448, 5, 11, device)
        calibrate_frozen_affine(tr.model, images.tensors)
        model = tr.model
        monkeypatch.setenv("CPM_EVAL_GRAPH", "0")
        thr = inference_leg(tr, images, device, forwards=2, rank_cut=80)["score_thresh"]
        model.eval()
        post = model.Grid_Cascade_RCNN.cls_post_processor
        post.score_thresh = thr                          # behind the 80-th class score of the untrained cls head
        a, b = images.tensors[0:1], images.tensors[1:2]
        with torch.no_grad():
            monkeypatch.setenv("CPM_EVAL_GRAPH", "0")
            eager_a, eager_b = _dets(model(a)), _dets(model(b))
            assert not model.__dict__.get("_eval_graphs")
            monkeypatch.setenv("CPM_EVAL_GRAPH", "1")
            first = _dets(model(a))                      # captures
            assert len(model._eval_graphs) == 1 and all(model._eval_graphs.values())
            again_a, again_b = _dets(model(a)), _dets(model(b))      # replays
            assert len(model._eval_graphs) == 1
            assert len(eager_a[0]) >= 5
            for got, want in ((first, eager_a), (again_a, eager_a), (again_b, eager_b)):
                for g, w in zip(got, want):
                    assert _same(g, w)
            assert not _same(eager_a[0], eager_b[0])
            # box_net (the test driver's entry: its caller keeps the features across augmentation passes) hands out
            # copies, not the graph's buffers: a second pass does not change what the first one returned
            model.Norm = torch.nn.Identity()             # (a model built for testing normalises here; this one is fed normalised images)
            monkeypatch.setenv("CPM_EVAL_GRAPH", "0")
            feats_e, res_e = model.box_net([a[0]])
            monkeypatch.setenv("CPM_EVAL_GRAPH", "1")
            n_graphs = len(model._eval_graphs)
            feats_a, res_a = model.box_net([a[0]])
            assert len(model._eval_graphs) == n_graphs + 1          # (a list of images is batched NCHW: another key)
            kept = [t.clone() for t in feats_a]
            feats_b, _ = model.box_net([b[0]])
            assert len(model._eval_graphs) == n_graphs + 1
            assert all(torch.equal(t, k) for t, k in zip(feats_a, kept))
            assert all(torch.equal(t, e) for t, e in zip(feats_a, feats_e))
            assert not torch.equal(feats_a[0], feats_b[0])
            for g, w in zip(_dets(res_a), _dets(res_e)):
                assert _same(g, w)
            del model.Norm
            # a parameter of the static part changes in place
            w = model.RPN.head.conv.weight
            w.mul_(1.5)                                  # (under no_grad: the version counter moves, as in copy_)
            w.add_(0.01)
            changed = _dets(model(a))
            monkeypatch.setenv("CPM_EVAL_GRAPH", "0")
            changed_eager = _dets(model(a))
            for g, e in zip(changed, changed_eager):
                assert _same(g, e)
            assert not _same(changed[0], eager_a[0])
    finally:
        _hip.set_deterministic(False)
        _hip.set_conv_math(prev)
        config.reset_cfg()
        torch.cuda.empty_cache()
