"""GPU: size-independent properties at BASELINE.json's full sizes (2 x 3x800x1333 -> 200x336 .. 13x21 maps, 268 569
anchors per image, 2000-box NMS segments, 192 grid RoIs): linearity and cross-arithmetic agreement of the convs,
NMS invariants, matcher invariants, and the grid target -> heat map -> box round trip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("shape", [(2, 256, 200, 336, 256, 3, 1, 1), (2, 1024, 50, 84, 256, 1, 1, 0),
                                   (2, 256, 100, 168, 128, 1, 2, 0)])
def test_conv_full_size_linearity_and_arithmetic_agreement(shape):
    """conv(a*x1 + b*x2) == a*conv(x1) + b*conv(x2) and bf16x3 == f32 arithmetic to 1e-4 of the tensor maximum, for
    forward, data gradient and weight gradient at the FPN-output / backbone shapes (the 3x3 case runs the patch
    kernel in bf16x3 mode and the generic implicit GEMM in f32 mode)."""
    from pet.lib.ops import _hip, conv as C
    n, c, h, w_, k, r, stride, pad = shape
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(n, c, h, w_, generator=g).cuda().contiguous(memory_format=CL)
    x2 = torch.randn(n, c, h, w_, generator=g).cuda().contiguous(memory_format=CL)
    w = (torch.randn(k, c, r, r, generator=g) * (2.0 / (c * r * r)) ** 0.5).cuda().contiguous(memory_format=CL)
    prev = _hip.get_conv_math()
    out = {}
    try:
        for mode in ("f32", "bf16x3"):
            _hip.set_conv_math(mode)
            f = lambda t: C.conv2d_forward(t, w, None, None, None, 0, False, stride, pad, 1, 1)
            y1, y2, y12 = f(x1), f(x2), f(0.5 * x1 - 2.0 * x2)
            assert _rel(y12, 0.5 * y1 - 2.0 * y2) < 1e-4
            dx = C.conv2d_backward_data(y1, w, tuple(x1.shape), stride, pad, 1, 1)
            dw = C.conv2d_backward_weight(x2, y1, w, stride, pad, 1, 1)
            out[mode] = (y1, dx, dw)
    finally:
        _hip.set_conv_math(prev)
    for a, b in zip(out["bf16x3"], out["f32"]):
        assert _rel(a, b) < 1e-4


def test_nms_full_size_invariants():
    """10 segments x 2000 boxes (the RPN shape): kept indices are in descending score order, no kept pair of a
    segment overlaps beyond the threshold, every dropped box overlaps a better kept one, NMS of the kept set keeps
    all of it."""
    import pet.lib.ops as ops
    g = torch.Generator().manual_seed(2)
    n_seg, per, thr = 10, 2000, 0.7
    xy = torch.rand(n_seg * per, 2, generator=g) * torch.tensor([1200., 700.])
    boxes = torch.cat([xy, xy + torch.rand(n_seg * per, 2, generator=g) * 250 + 8], 1).cuda()
    scores = torch.rand(n_seg * per, generator=g).cuda()
    offs = [i * per for i in range(n_seg + 1)]
    keep, counts = ops.nms_segments(boxes, scores, None, offs, thr, 0)
    counts = counts.tolist()
    for s in range(n_seg):
        k = keep[offs[s]:offs[s] + counts[s]] + offs[s]
        sc = scores[k]
        assert bool((sc[:-1] >= sc[1:]).all())
        iou = ops.box_iou(boxes[k], boxes[k])
        iou.fill_diagonal_(0)
        assert float(iou.max()) <= thr
        mask = torch.ones(per, dtype=torch.bool, device="cuda")
        mask[k - offs[s]] = False
        dropped = torch.nonzero(mask).squeeze(1) + offs[s]
        cover = ops.box_iou(boxes[dropped], boxes[k])                  # [dropped, kept]
        better = scores[k][None, :] >= scores[dropped][:, None]
        assert bool(((cover > thr) & better).any(dim=1).all())
        k2, c2 = ops.nms_segments(boxes[k].contiguous(), sc.contiguous(), None, [0, counts[s]], thr, 0)
        assert int(c2[0]) == counts[s]


def test_match_rois_full_size_invariants():
    """All 268 569 anchors of two images against 16 gts each: value ranges, threshold consistency, agreement with the
    per-image tensor-op Matcher on a random subset, low-quality matches (every gt keeps its best anchors)."""
    import pet.lib.ops as ops
    from pet.rcnn.core import config
    from pet.rcnn.modeling.rpn.anchor_generator import make_anchor_generator
    from pet.rcnn.utils.matcher import Matcher
    from pet.utils.data.structures.boxlist_ops import box_iou_plus1
    from test_host_logic import CPM_OPTS
    config.reset_cfg()
    config.merge_cfg_from_list(CPM_OPTS)
    try:
        ag = make_anchor_generator()
        anchors = torch.cat(ag.grid_anchors([(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]), 0).cuda()
        assert anchors.shape[0] == 268569
        g = torch.Generator().manual_seed(3)
        gts = []
        for _ in range(2):
            xy = torch.rand(16, 2, generator=g) * torch.tensor([900., 500.])
            gts.append(torch.cat([xy, xy + torch.rand(16, 2, generator=g) * 368 + 32], 1).cuda())
        rois = torch.cat([anchors, anchors], 0)
        img = torch.arange(2, device="cuda").repeat_interleave(anchors.shape[0]).to(torch.int32)
        off = torch.tensor([0, 16, 32], dtype=torch.int32, device="cuda")
        m, v = ops.match_rois(rois, img, torch.cat(gts), off, 0.7, 0.3, True)
        assert int(m.min()) >= -2 and int(m.max()) < 16
        assert bool((m[v >= 0.7] >= 0).all()) and bool((m[(v >= 0.3) & (v < 0.7) & (m < 0)] == -2).all())
        for i in range(2):
            sl = slice(i * anchors.shape[0], (i + 1) * anchors.shape[0])
            q = box_iou_plus1(gts[i], anchors)
            want = Matcher(0.7, 0.3, True)(q)
            assert torch.equal(m[sl], want)
            best_per_gt = q.max(dim=1)[0]
            for j in range(16):                                    # each gt's best anchors are matched (to some gt)
                tied = torch.nonzero(q[j] == best_per_gt[j]).squeeze(1)
                assert bool((m[sl][tied] >= 0).all())
    finally:
        config.reset_cfg()


@pytest.mark.parametrize("ratio", [1.0, 0.5, 0.25])
def test_grid_target_decode_round_trip_full_size(oracle, ratio):
    """192 RoIs (MAX_SAMPLE_NUM_GRID x 2): rasterise the 9 point targets of each RoI for gt == RoI (C oracle), turn them
    into confident logits, decode on the device: every recovered coordinate lies within two heat-map cells of the box (the radius-1 target
    saturates a small patch and argmax returns its first cell in row-major order, up to one cell up/left of the centre;
    a second cell covers the target's own quantisation)."""
    import pet.lib.ops as ops
    from pet.rcnn.modeling.grid_rcnn.loss import calc_sub_regions
    rng = np.random.default_rng(4)
    R = 192
    xy = rng.uniform(0, [1000, 600], (R, 2))
    boxes = np.concatenate([xy, xy + rng.uniform(40, 320, (R, 2))], 1).astype(np.float32)
    tgt = oracle.grid_targets(boxes, boxes, 9, 56, 1, ratio)                     # [R, 9, 28, 28] 0/1
    assert tgt.sum(axis=(2, 3)).min() >= 1
    logits = torch.from_numpy(tgt * 20 - 10).cuda().contiguous(memory_format=CL)
    got = ops.grid_decode(logits, torch.from_numpy(boxes).cuda(), 56, calc_sub_regions(9, 3, 56), ratio).cpu().numpy()
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cell_w, cell_h = w * (1 + ratio) / 56, h * (1 + ratio) / 56
    assert np.all(np.abs(got[:, 0] - boxes[:, 0]) <= 2.0 * cell_w) and np.all(np.abs(got[:, 2] - boxes[:, 2]) <= 2.0 * cell_w)
    assert np.all(np.abs(got[:, 1] - boxes[:, 1]) <= 2.0 * cell_h) and np.all(np.abs(got[:, 3] - boxes[:, 3]) <= 2.0 * cell_h)
