"""GPU: the RPN head's sparse backward (csrc/rpn_sparse.hip, ops.rpn_head) against the dense formulation.

The RPN loss sums over the sampled anchors only (rpn/loss.py:88-126, 256 per image), so the gradient entering the head
(rpn/rpn.py:34-41) is exactly zero at every other anchor; the sparse backward computes the same gradients from the
sampled anchors' rows alone.  Held here: both RPN losses equal, the gradients of the five feature maps and of the six
head parameters equal to the dense ones up to summation order -- on the head alone (leaf features) and through the whole
training step (shared gradient accumulators of the FPN maps, flat-optimizer-free autograd gradients)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def relmax(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.fixture(params=[2, 1], ids=["two_images", "one_image"])
def trainer(request):
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    from bench import Trainer, calibrate_frozen_affine, synthetic_batch
    from pet.lib.ops import _hip
    from pet.rcnn.core import config
    prev = _hip.get_conv_math()
    _hip.set_conv_math("bf16x3")
    dev = torch.device("cuda", 0)
    tr = Trainer(dev)
    # (one image: the gradient slices the loss hands back carry a batch stride that means nothing)
    images, targets = synthetic_batch(request.param, 320, 448, 6, 21, dev)
    calibrate_frozen_affine(tr.model, images.tensors)
    yield tr, images, targets
    _hip.set_conv_math(prev)
    config.reset_cfg()


def test_mask_compact_lists_the_flagged_positions_in_order():
    import pet.lib.ops as ops
    g = torch.Generator().manual_seed(0)
    for total, npos in ((5000, 37), (537138, 512), (100, 0), (3000, 3000)):
        m = torch.zeros(total, dtype=torch.bool)
        m[torch.randperm(total, generator=g)[:npos]] = True
        pos = (m & (torch.rand(total, generator=g) < 0.5)).cuda()
        neg = (m & ~pos.cpu()).cuda()
        cap = max(npos, 1) + 5
        idx, cnt = ops.mask_compact(pos, neg, cap)
        want = torch.nonzero(m).squeeze(1).to(torch.int32)
        assert int(cnt) == npos
        assert torch.equal(idx[:npos].cpu(), want) and bool((idx[npos:] == -1).all())


@pytest.mark.parametrize("leaf", [True, False], ids=["head_alone", "whole_step"])
def test_sparse_rpn_backward_equals_dense(trainer, leaf, deterministic_reductions):
    """(ordered reductions everywhere else: the forward pass, and with it the proposals and every RoI set of the step,
    is then the same computation in both runs; the sparse path itself is forced on with _RPN_SPARSE = 2)"""
    from pet.lib.ops import conv as C
    tr, images, targets = trainer
    model = tr.model
    model.train()
    params = {k: p for k, p in model.named_parameters() if p.requires_grad}

    def run(sparse):
        C._RPN_SPARSE = 2 if sparse else 0
        torch.manual_seed(7)                              # the samplers draw their seeds from torch's CPU generator
        tr.optimizer.zero_grad()
        if leaf:
            with torch.no_grad():
                feats = model._features(images.tensors)
            feats = [f.detach().clone().contiguous(memory_format=CL).requires_grad_(True) for f in feats]
            _, losses = model.RPN(images, feats, targets)
            (losses["loss_objectness"] + losses["loss_rpn_box_reg"]).backward()
            grads = {"feat%d" % i: f.grad.detach().clone() for i, f in enumerate(feats)}
            keys = [k for k in params if k.startswith("RPN.head.")]
        else:
            out = model(images, targets)
            losses = out["losses"]
            torch.autograd.backward([v for v in losses.values() if v.requires_grad])
            grads = {}
            keys = list(params)
        torch.cuda.synchronize()
        for k in keys:
            grads[k] = params[k].grad.detach().clone()
        return {k: float(v.detach()) for k, v in losses.items()}, grads
    try:
        l0, g0 = run(False)
        l1, g1 = run(True)
    finally:
        C._RPN_SPARSE = 1
    for k in ("loss_objectness", "loss_rpn_box_reg"):
        assert abs(l0[k] - l1[k]) <= 1e-6 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    assert len(g0) >= 11 and set(g0) == set(g1)
    head = [k for k in g0 if k.startswith("RPN.head.")]
    assert len(head) == 6
    for k in head:
        assert float(g0[k].abs().max()) > 0, k
    worst = {k: relmax(g1[k], g0[k]) for k in g0}
    # summation order only (the sparse path adds <= 512 rows, the dense one every pixel; float atomics in the scatter);
    # through the whole step the RoI samplers are seeded alike, so every other gradient is the same computation
    tol = 2e-5 if leaf else 2e-4
    bad = {k: v for k, v in worst.items() if not v < tol}
    assert not bad, bad
