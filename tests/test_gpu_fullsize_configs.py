"""GPU: every BASELINE training configuration at FULL depth and FULL size:

  #2  R-50-FPN CPM R-CNN             LAYERS (3, 4, 6, 3), bs = 2, 3 x 800 x 1333 (the headline configuration)
  #4  R-101-FPN CPM R-CNN            LAYERS (3, 4, 23, 3), bs = 2, 3 x 800 x 1333
  #5  X-101-64x4d-FPN + DCN           ResNeXt (3, 4, 23, 3), C = 64, width 4, deformable conv in layer2-4,
                                      MAX_SAMPLE_NUM_GRID 32, bs = 1, 3 x 800 x 1333

One whole training iteration each through the same loop as bench.py (scheduler, zero_grad, forward, backward, SGD):
state-dict ABI equal to the reference's dump, finite losses, every trainable tensor of the reference's trainable set
receives a finite, non-zero gradient, RoI counts inside the configuration's caps, and the step is repeatable.
Numerical parity at this size: tests/test_gpu_fullsize_oracle.py (config #2 against the CPU oracle); the block types of
#4 / #5 are held at small sizes (test_gpu_model.py: R-50 blocks vs the reference; test_gpu_deform.py: the ResNeXt-DCN
body vs the CPU oracle)."""
import json
import os

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(body, layers, batch, meta_file, grid_cap):
    from bench import Trainer, calibrate_frozen_affine, synthetic_batch
    from pet.rcnn.core import config
    device = torch.device("cuda", 0)
    with open(os.path.join(ROOT, "tests", "golden", meta_file)) as f:
        meta = json.load(f)
    try:
        tr = Trainer(device, layers=layers, body=body)
        tr.optimizer.clear_grads_in_step = False       # this test reads the gradients behind the step (bench.py's loop
        model = tr.model                                # lets the SGD kernel clear them: FlatSGD.clear_grads_in_step)
        assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == meta["state_dict"]
        trainable = [k for k, p in model.named_parameters() if p.requires_grad]
        assert trainable == meta["trainable"]
        images, targets = synthetic_batch(batch, 800, 1333, 16, 1234, device)
        cal, _ = synthetic_batch(batch, 800, 1333, 1, 4321, device)
        calibrate_frozen_affine(model, cal.tensors)
        assert tuple(images.tensors.shape) == (batch, 3, 800, 1344)
        losses = []
        for it in range(2):
            tr.step(images, targets)
            loss = sum(v.detach() for v in tr.last_losses.values())
            torch.cuda.synchronize()
            assert torch.isfinite(loss), (it, {k: float(v) for k, v in tr.last_losses.items()})
            assert set(tr.last_losses) == {"loss_objectness", "loss_rpn_box_reg", "loss_classifier", "loss_grid_1",
                                           "loss_grid_2", "loss_grid_3", "loss_iou_3", "loss_rescore"}
            losses.append(float(loss.detach()))
            if it == 0:
                params = dict(model.named_parameters())
                n = 0
                for k in trainable:
                    g = params[k].grad
                    assert g is not None and bool(torch.isfinite(g).all()), k
                    n += int(float(g.abs().max()) > 0)
                # zero-initialised tensors whose inputs' gradient is identically zero at step 0 may stay zero
                # (DeformConvPack offset predictors get a gradient, their zero weights do not block it)
                assert n >= len(trainable) - 2, (n, len(trainable))
        counts = model.Grid_Cascade_RCNN.last_counts
        assert 0 < counts["cls"] <= 512 * batch and 0 < counts["rescore"] <= 512 * batch
        for s in range(3):
            assert 0 < counts["grid_%d" % s] <= grid_cap * batch + 16 * batch, counts
        return losses, counts
    finally:
        config.reset_cfg()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("math", ["bf16x3", "f32"])
def test_r50_full_size_training_step(math):
    from pet.lib.ops import _hip
    prev = _hip.get_conv_math()
    _hip.set_conv_math(math)
    try:
        losses, counts = _run("resnet", (3, 4, 6, 3), 2, "model_r50_meta.json", 96)
    finally:
        _hip.set_conv_math(prev)
    assert len(losses) == 2


@pytest.mark.parametrize("math", ["bf16x3"])
def test_r101_full_size_training_step(math):
    from pet.lib.ops import _hip
    prev = _hip.get_conv_math()
    _hip.set_conv_math(math)
    try:
        losses, counts = _run("resnet", (3, 4, 23, 3), 2, "model_r101_meta.json", 96)
    finally:
        _hip.set_conv_math(prev)
    assert len(losses) == 2


@pytest.mark.parametrize("math", ["bf16x3"])
def test_x101_dcn_full_size_training_step(math):
    from pet.lib.ops import _hip
    prev = _hip.get_conv_math()
    _hip.set_conv_math(math)
    try:
        losses, counts = _run("x101dcn", (3, 4, 23, 3), 1, "model_x101_meta.json", 32)
    finally:
        _hip.set_conv_math(prev)
    assert len(losses) == 2


def test_x101_body_forward_repeats_with_side_sections(deterministic_reductions):
    """The forward side sections (ops.fwd_fork / fwd_side / fwd_join) queue a branch's kernels on the second stream,
    forked from the compute stream some way back.  A tensor such an op creates must not come out of the compute
    stream's allocator pool: that pool may hand out a block whose previous owner -- a bottleneck's conv1 output, dropped
    when conv2 returned -- compute-stream kernels behind the fork point are still reading (round 5: the X-101 body's
    layer1.0 read garbage whenever the pool offered that block; R-50 never hit it).  H.side_alloc takes them from the
    second stream's pool.  Held here: the X-101-64x4d-DCN body at full size (bs = 1), forward repeated without and with
    autograd -- immediate frees in the first two -- gives the SAME C2..C5 every time (ordered reductions: bit for bit),
    equal to the run with the side sections switched off."""
    import __graft_entry__ as entry
    entry.ensure_built()
    from bench import Trainer, calibrate_frozen_affine, synthetic_batch
    from pet.lib.ops import _hip, conv as C
    from pet.rcnn.core import config
    device = torch.device("cuda", 0)
    prev = _hip.get_conv_math()
    _hip.set_conv_math("bf16x3")
    try:
        tr = Trainer(device, body="x101dcn", hold_offsets=True)
        images, _ = synthetic_batch(1, 800, 1333, 16, 1234, device)
        cal, _ = synthetic_batch(1, 800, 1333, 1, 4321, device)
        calibrate_frozen_affine(tr.model, cal.tensors)
        body = tr.model.Conv_Body
        assert C._FWD_SIDE and C._SIDE_WGRAD
        runs = []
        for rep in range(4):
            with torch.set_grad_enabled(rep >= 2):
                runs.append([t.detach().clone() for t in body(images.tensors)])
            torch.cuda.synchronize()
        C._FWD_SIDE = False
        try:
            with torch.no_grad():
                plain = [t.detach().clone() for t in body(images.tensors)]
        finally:
            C._FWD_SIDE = True
        for rep, outs in enumerate(runs):
            for lvl, (a, b) in enumerate(zip(outs, plain)):
                assert torch.equal(a, b), "repetition %d, C%d: max diff %g of %g" % (
                    rep, lvl + 2, float((a - b).abs().max()), float(b.abs().max()))
    finally:
        _hip.set_conv_math(prev)
        config.reset_cfg()
        torch.cuda.empty_cache()
