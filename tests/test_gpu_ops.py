"""GPU parity tests (run with -m gpu on the MI355X box): HIP RoIAlign / NMS / box ops through the
C ABI (pet.lib.ops._C -> libcpmrcnn_hip.so) against the CPU oracle and the reference goldens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def C():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    from pet.lib.ops import _C
    return _C


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


@pytest.mark.parametrize("ph", [7, 14])
@pytest.mark.parametrize("inv_scale", [4, 8, 16, 32])
@pytest.mark.parametrize("nhwc", [False, True])
def test_roi_align_golden(golden_ops, C, ph, inv_scale, nhwc):
    g = golden_ops
    key = "roi_%d_%d" % (ph, inv_scale)
    x, rois = dev(g[key + "_x"]), dev(g[key + "_rois"])
    B, Cc, H, W = x.shape
    if nhwc:
        x = x.contiguous(memory_format=torch.channels_last)
    for interp in (0, 1):
        for aligned in (0, 1):
            sfx = "_i%d_a%d" % (interp, aligned)
            y = C.roi_align_forward(x, rois, 1.0 / inv_scale, ph, ph, 2, bool(aligned), interp)
            assert y.shape == (rois.shape[0], Cc, ph, ph)
            assert np.array_equal(y.cpu().numpy(), g[key + sfx + "_y"]), "forward must be bit-exact"
            go = dev(g[key + sfx + "_g"])
            if nhwc:
                go = go.contiguous(memory_format=torch.channels_last)
            gi = C.roi_align_backward(go, rois, 1.0 / inv_scale, ph, ph, B, Cc, H, W, 2, bool(aligned), interp)
            # atomic accumulation order differs from the serial CPU loop (hundreds of terms of mixed sign
            # land on each pixel of these tiny maps): tolerance 1e-4, inside north_star's 1e-3 fp32
            np.testing.assert_allclose(gi.cpu().numpy(), g[key + sfx + "_gi"], rtol=1e-4, atol=1e-4)


def test_roi_align_adaptive_and_empty(golden_ops, C):
    g = golden_ops
    y = C.roi_align_forward(dev(g["roi_adapt_x"]), dev(g["roi_adapt_rois"]), 0.25, 7, 7, 0, False, 0)
    assert np.array_equal(y.cpu().numpy(), g["roi_adapt_y"])
    e = C.roi_align_forward(dev(g["roi_adapt_x"]), torch.zeros((0, 5), device="cuda"), 0.25, 7, 7, 2, False, 0)
    assert e.shape == (0, 3, 7, 7)
    with pytest.raises(RuntimeError):
        C.roi_align_forward(dev(g["roi_adapt_x"]), dev(g["roi_adapt_rois"]), 0.25, 7, 7, 2, False, 3)
    with pytest.raises(RuntimeError):
        C.roi_align_forward(torch.zeros(1, 3, 8, 8), torch.zeros(1, 5), 0.25, 7, 7, 2, False, 0)   # CPU tensors


def _random_rois(rng, K, B, iw, ih):
    b = rng.integers(0, B, K).astype(np.float32)
    w = np.exp(rng.uniform(np.log(4), np.log(iw), K))
    h = np.exp(rng.uniform(np.log(4), np.log(ih), K))
    x1 = rng.uniform(-10, iw - 4, K)
    y1 = rng.uniform(-10, ih - 4, K)
    return np.stack([b, x1, y1, x1 + w, y1 + h], 1).astype(np.float32)


def test_roi_align_fpn_fused_vs_oracle(oracle, C):
    """Fused multi-level pooler == LevelMapper + per-level RoIAlign + scatter (poolers.py:90-132)."""
    from pet.lib.ops import _hip as H
    from pet.lib.ops.pooler_fpn import roi_align_fpn
    rng = np.random.default_rng(11)
    B, Cc = 2, 64
    sizes = [(48, 80), (24, 40), (12, 20), (6, 10)]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    feats = [rng.standard_normal((B, Cc, h, w)).astype(np.float32) for h, w in sizes]
    rois = _random_rois(rng, 300, B, 320, 192)
    rois[:4, 1:] = [[0, 0, 55, 55], [0, 0, 111, 111], [0, 0, 223, 223], [0, 0, 447, 447]]   # level boundaries
    lv = oracle.level_map(rois[:, 1:])
    for ph in (7, 14):
        want = np.zeros((rois.shape[0], Cc, ph, ph), np.float32)
        for l in range(4):
            idx = np.nonzero(lv == l)[0]
            if len(idx):
                want[idx] = oracle.roi_align_forward(feats[l], rois[idx], scales[l], ph, ph, 2, False, 0)
        tf = [dev(f).contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats]
        y, levels = roi_align_fpn(tf, dev(rois), (ph, ph), scales, 2, return_levels=True)
        assert np.array_equal(levels.cpu().numpy().astype(np.int64), lv), "RoI->level indices must be bit-exact"
        assert np.array_equal(y.detach().cpu().numpy(), want)
        go = rng.standard_normal(want.shape).astype(np.float32)
        y.backward(dev(go).contiguous(memory_format=torch.channels_last))
        for l in range(4):
            idx = np.nonzero(lv == l)[0]
            wg = oracle.roi_align_backward(go[idx], rois[idx], scales[l], ph, ph, B, Cc, sizes[l][0], sizes[l][1], 2)
            np.testing.assert_allclose(tf[l].grad.cpu().numpy(), wg, rtol=1e-4, atol=1e-4)


def test_roi_align_full_size_property(C):
    """BASELINE-size case (K=1024, C=256, 7x7 on the 200x336 level): constant map -> constant output,
    linearity in the input, and sum(grad_input) == sum(grad_output) for in-bounds RoIs."""
    torch.manual_seed(0)
    B, Cc, H, W = 2, 256, 200, 336
    rng = np.random.default_rng(1)
    rois = _random_rois(rng, 1024, B, 1300, 780)
    rois[:, 1:3] = np.abs(rois[:, 1:3]) + 8
    rois[:, 3] = np.minimum(rois[:, 3], 1300)
    rois[:, 4] = np.minimum(rois[:, 4], 780)
    r = dev(rois)
    ones = torch.full((B, Cc, H, W), 3.25, device="cuda").contiguous(memory_format=torch.channels_last)
    y = C.roi_align_forward(ones, r, 0.25, 7, 7, 2, False, 0)
    torch.testing.assert_close(y, torch.full_like(y, 3.25), rtol=1e-6, atol=0)
    a = torch.randn(B, Cc, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    b = torch.randn(B, Cc, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    ya, yb = C.roi_align_forward(a, r, 0.25, 7, 7, 2, False, 0), C.roi_align_forward(b, r, 0.25, 7, 7, 2, False, 0)
    yab = C.roi_align_forward(a + b, r, 0.25, 7, 7, 2, False, 0)
    torch.testing.assert_close(yab, ya + yb, rtol=1e-5, atol=1e-5)
    # NCHW kernel agrees with the NHWC kernel bit for bit
    yn = C.roi_align_forward(a.contiguous(), r, 0.25, 7, 7, 2, False, 0)
    assert torch.equal(yn, ya.contiguous())
    go = torch.randn_like(ya)
    gi = C.roi_align_backward(go, r, 0.25, 7, 7, B, Cc, H, W, 2, False, 0)
    torch.testing.assert_close(gi.double().sum(), go.double().sum(), rtol=1e-4, atol=1e-2)


def _rand_boxes(rng, n, span=400):
    xy = rng.uniform(0, span, (n, 2))
    wh = rng.uniform(4, 120, (n, 2))
    return np.concatenate([xy, xy + wh], 1).astype(np.float32)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 700, 2000, 5000])
def test_nms_vs_oracle(oracle, C, n):
    rng = np.random.default_rng(n)
    boxes = _rand_boxes(rng, n)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    if n > 8:
        scores[: n // 5] = scores[0]            # ties -> lower index first
        boxes[n // 2] = boxes[0]                # exact duplicate
    for thr in (0.3, 0.7):
        got = C.nms(dev(boxes), dev(scores), thr).cpu().numpy()
        assert np.array_equal(got, oracle.nms(boxes, scores, thr)), "keep set must be bit-exact"
    labels = rng.integers(1, 6, n).astype(np.int64)
    got = C.ml_nms(dev(boxes), dev(scores), dev(labels, torch.int64), 0.3, 0).cpu().numpy()
    assert np.array_equal(got, oracle.ml_nms(boxes, scores, labels, 0.3))
    got = C.ml_nms(dev(boxes), dev(scores), dev(labels, torch.int64), 0.3, 5).cpu().numpy()
    assert np.array_equal(got, oracle.ml_nms(boxes, scores, labels, 0.3, 5))


def test_nms_empty_and_errors(C):
    e = C.nms(torch.zeros((0, 4), device="cuda"), torch.zeros((0,), device="cuda"), 0.5)
    assert e.dtype == torch.int64 and e.numel() == 0 and e.is_cuda
    e = C.ml_nms(torch.zeros((0, 4), device="cuda"), torch.zeros((0,), device="cuda"),
                 torch.zeros((0,), dtype=torch.int64, device="cuda"), 0.5, 0)
    assert e.numel() == 0
    with pytest.raises(RuntimeError):
        C.nms(torch.zeros((4, 4)), torch.zeros((4,)), 0.5)


def test_nms_segments_rpn_shape(oracle, C):
    """The RPN call shape: image x level segments of <= 2000 pre-sorted candidates (rpn/inference.py:67-114)."""
    rng = np.random.default_rng(5)
    sizes = [2000, 2000, 2000, 1575, 273, 2000, 2000, 2000, 1575, 273, 0, 7]
    off = np.concatenate([[0], np.cumsum(sizes)])
    boxes = _rand_boxes(rng, off[-1], span=1200)
    scores = rng.uniform(0, 1, off[-1]).astype(np.float32)
    keep, counts = C.nms_segments(dev(boxes), dev(scores), None, off.tolist(), 0.7, 0)
    keep, counts = keep.cpu().numpy(), counts.cpu().numpy()
    for p, n in enumerate(sizes):
        want = oracle.nms(boxes[off[p]:off[p + 1]], scores[off[p]:off[p + 1]], 0.7)
        assert counts[p] == len(want)
        assert np.array_equal(keep[off[p]: off[p] + counts[p]], want)


def test_nms_segments_presorted_equals_sorting_path(oracle, C):
    """Segments whose scores already descend (rows of a sorted top-k, ties included): the entry that skips sort and
    gather keeps the same boxes in the same order as the sorting one, and as the oracle."""
    rng = np.random.default_rng(6)
    sizes = [2000, 2048, 1, 65, 0, 1575, 273]
    off = np.concatenate([[0], np.cumsum(sizes)])
    boxes = _rand_boxes(rng, off[-1], span=900)
    scores = np.concatenate([-np.sort(-np.round(rng.uniform(0, 1, n), 2)) for n in sizes]).astype(np.float32)   # ties
    a, ca = C.nms_segments(dev(boxes), dev(scores), None, off.tolist(), 0.7, 0)
    b, cb = C.nms_segments(dev(boxes), dev(scores), None, off.tolist(), 0.7, 0, presorted=True)
    assert torch.equal(ca, cb)
    ca = ca.cpu().numpy()
    for p, n in enumerate(sizes):
        assert torch.equal(a[off[p]: off[p] + ca[p]], b[off[p]: off[p] + ca[p]])
        want = oracle.nms(boxes[off[p]:off[p + 1]], scores[off[p]:off[p + 1]], 0.7)
        assert np.array_equal(b[off[p]: off[p] + ca[p]].cpu().numpy(), want)
    c, cc = C.nms_segments(dev(boxes), dev(scores), None, off.tolist(), 0.7, 100, presorted=True)        # topk early-out
    d, cd = C.nms_segments(dev(boxes), dev(scores), None, off.tolist(), 0.7, 100)
    assert torch.equal(cc, cd)
    for p in range(len(sizes)):
        assert torch.equal(c[off[p]: off[p] + int(cc[p])], d[off[p]: off[p] + int(cc[p])])


def test_nms_big_segment_global_sort(oracle, C):
    rng = np.random.default_rng(9)
    n = 20000                                   # > 16384: global bitonic path
    boxes = _rand_boxes(rng, n, span=3000)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    labels = rng.integers(1, 81, n).astype(np.int64)
    got = C.ml_nms(dev(boxes), dev(scores), dev(labels, torch.int64), 0.3, 0).cpu().numpy()
    assert np.array_equal(got, oracle.ml_nms(boxes, scores, labels, 0.3))
    # properties at this size: kept scores are sorted, no kept same-label pair overlaps above the threshold
    ks = scores[got]
    assert np.all(ks[:-1] >= ks[1:])
    sub = got[:400]
    iou = oracle.box_iou(boxes[sub], boxes[sub])
    same = labels[sub][:, None] == labels[sub][None, :]
    np.fill_diagonal(iou, 0)
    assert not np.any((iou > 0.3) & same)


def test_box_iou_and_pool_points(oracle, C):
    rng = np.random.default_rng(2)
    a, b = _rand_boxes(rng, 300), _rand_boxes(rng, 77)
    assert np.array_equal(C.box_iou(dev(a), dev(b)).cpu().numpy(), oracle.box_iou(a, b))
    x = rng.standard_normal((2, 5, 20, 30)).astype(np.float32)
    K = 2 * 196
    pts = np.stack([np.zeros(K), rng.uniform(-8, 130, K), rng.uniform(-8, 90, K)], 1).astype(np.float32)
    y = C.pool_points_interp_forward(dev(x), dev(pts), 0.25)
    assert np.array_equal(y.cpu().numpy(), oracle.pool_points_interp_forward(x, pts, 0.25))
    go = rng.standard_normal((K, 5)).astype(np.float32)
    gi = C.pool_points_interp_backward(dev(go), dev(pts), 0.25, 2, 5, 20, 30)
    np.testing.assert_allclose(gi.cpu().numpy(), oracle.pool_points_interp_backward(go, pts, 0.25, 2, 5, 20, 30),
                               rtol=1e-5, atol=1e-5)


def _fpn_backward(kind, go, rois, shapes, scales, ph, ratio, init=None):
    """Run one of the two multi-level backward kernels straight through the C ABI."""
    import ctypes
    from pet.lib.ops import _hip as H
    n = len(shapes)
    grads = []
    for i, s in enumerate(shapes):
        if init is not None and init[i] is not None:
            grads.append(init[i].clone())
        elif kind == "atomic":
            grads.append(torch.zeros(s, device="cuda").contiguous(memory_format=torch.channels_last))
        else:
            grads.append(torch.full(s, float("nan"), device="cuda").contiguous(memory_format=torch.channels_last))
    hs = (ctypes.c_int * n)(*[s[2] for s in shapes])
    ws = (ctypes.c_int * n)(*[s[3] for s in shapes])
    sc = (ctypes.c_float * n)(*scales)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in grads])
    K, B, Cc = rois.shape[0], shapes[0][0], shapes[0][1]
    common = (H.ptr(go), ptrs, hs, ws, sc, n, H.ptr(rois), K, B, Cc, ph, ph, ratio, H.f(2.0), H.f(5.0), H.f(224.0),
              H.f(4.0), H.f(1e-6))
    if kind == "atomic":
        rc = H.lib().cpm_roi_align_fpn_backward(*common, H.stream())
    else:
        need = H.lib().cpm_roi_align_fpn_gather_workspace_bytes(hs, ws, n, B, K)
        wsb = torch.empty(max(int(need), 1), dtype=torch.uint8, device="cuda")
        mask = 0 if init is None else sum((1 if t is not None else 0) << i for i, t in enumerate(init))
        rc = H.lib().cpm_roi_align_fpn_backward_gather(*common, mask, H.ptr(wsb), H.c_size_t(wsb.numel()), H.stream())
    H.check(rc, "fpn backward " + kind)
    torch.cuda.synchronize()
    return grads


@pytest.mark.parametrize("ph,ratio", [(7, 2), (14, 2), (7, 0), (3, 1)])
def test_roi_align_fpn_gather_backward(oracle, ph, ratio):
    """The atomic-free gather formulation of the multi-level backward: equals the oracle (hence the scatter kernel),
    overwrites un-initialised maps completely, adds into the levels flagged for accumulation, is bit-reproducible, and
    handles sub-pixel, border-straddling, out-of-image and empty inputs."""
    rng = np.random.default_rng(ph * 10 + ratio)
    B, Cc = 2, 64
    sizes = [(48, 80), (24, 40), (12, 20), (6, 10)]
    shapes = [(B, Cc, h, w) for h, w in sizes]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    rois = _random_rois(rng, 300, B, 320, 192)
    rois[:8, 1:] = [[0, 0, 55, 55], [0, 0, 111, 111], [0, 0, 223, 223], [0, 0, 447, 447], [100.2, 50.1, 100.9, 50.7],
                    [-40, -30, 20, 10], [300, 170, 400, 260], [-500, -500, -400, -400]]
    lv = oracle.level_map(rois[:, 1:])
    go = rng.standard_normal((rois.shape[0], Cc, ph, ph)).astype(np.float32)
    g_dev = dev(go).contiguous(memory_format=torch.channels_last)
    r_dev = dev(rois)
    got = _fpn_backward("gather", g_dev, r_dev, shapes, scales, ph, ratio)
    for l in range(4):
        idx = np.nonzero(lv == l)[0]
        want = oracle.roi_align_backward(go[idx], rois[idx], scales[l], ph, ph, B, Cc, sizes[l][0], sizes[l][1], ratio)
        assert torch.isfinite(got[l]).all()                          # every element of the NaN-filled map was written
        np.testing.assert_allclose(got[l].cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    atom = _fpn_backward("atomic", g_dev, r_dev, shapes, scales, ph, ratio)
    for a, b in zip(got, atom):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    again = _fpn_backward("gather", g_dev, r_dev, shapes, scales, ph, ratio)
    for a, b in zip(got, again):
        assert torch.equal(a, b)                                      # no atomics: same bits every run
    base = [torch.randn(s, device="cuda").contiguous(memory_format=torch.channels_last) if i % 2 else None
            for i, s in enumerate(shapes)]
    mixed = _fpn_backward("gather", g_dev, r_dev, shapes, scales, ph, ratio, init=base)
    for i in range(4):
        torch.testing.assert_close(mixed[i], got[i] + base[i] if base[i] is not None else got[i], rtol=1e-5, atol=1e-5)
    empty = _fpn_backward("gather", g_dev[:0], r_dev[:0], shapes, scales, ph, ratio)
    assert all(bool((t == 0).all()) for t in empty)


def test_roi_align_fpn_gather_backward_several_sets():
    """cpm_roi_align_fpn_backward_gather_sets: the pooled gradients of several heads (own RoIs, 7x7 / 14x14 / 3x3 bins,
    own sampling ratios, an empty set among them) in one pass == the sum of one call per set, into maps that already
    hold something; bit-reproducible."""
    import ctypes
    from pet.lib.ops import _hip as H
    rng = np.random.default_rng(77)
    B, Cc = 2, 64
    sizes = [(48, 80), (24, 40), (12, 20), (6, 10)]
    shapes = [(B, Cc, h, w) for h, w in sizes]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    sets = [(300, 7, 2), (40, 14, 2), (0, 7, 2), (25, 14, 2), (200, 3, 0)]          # (K, pooled size, sampling ratio)
    rois = [dev(_random_rois(rng, max(k, 1), B, 320, 192)[:k]) for k, _, _ in sets]
    gos = [torch.randn(k, Cc, ph, ph, device="cuda").contiguous(memory_format=torch.channels_last) for k, ph, _ in sets]
    base = [torch.randn(s, device="cuda").contiguous(memory_format=torch.channels_last) for s in shapes]
    want = [t.clone() for t in base]
    for (k, ph, ratio), r, go in zip(sets, rois, gos):
        if k:
            want = _fpn_backward("gather", go, r, shapes, scales, ph, ratio, init=want)

    def together():
        acc = [t.clone() for t in base]
        n, m = len(shapes), len(sets)
        hs = (ctypes.c_int * n)(*[s_[2] for s_ in shapes])
        ws = (ctypes.c_int * n)(*[s_[3] for s_ in shapes])
        sc = (ctypes.c_float * n)(*scales)
        vp, ip = ctypes.c_void_p * m, ctypes.c_int * m
        ktot = sum(k for k, _, _ in sets)
        need = H.lib().cpm_roi_align_fpn_gather_workspace_bytes(hs, ws, n, B, ktot)
        wsb = torch.empty(int(need), dtype=torch.uint8, device="cuda")
        rc = H.lib().cpm_roi_align_fpn_backward_gather_sets(
            m, vp(*[g.data_ptr() if g.numel() else None for g in gos]),
            vp(*[r.data_ptr() if r.numel() else None for r in rois]), ip(*[k for k, _, _ in sets]),
            ip(*[p for _, p, _ in sets]), ip(*[p for _, p, _ in sets]), ip(*[q for _, _, q in sets]),
            (ctypes.c_void_p * n)(*[t.data_ptr() for t in acc]), hs, ws, sc, n, B, Cc, H.f(2.0), H.f(5.0), H.f(224.0),
            H.f(4.0), H.f(1e-6), (1 << n) - 1, H.ptr(wsb), H.c_size_t(wsb.numel()), H.stream())
        H.check(rc, "gather_sets")
        torch.cuda.synchronize()
        return acc
    got, again = together(), together()
    for a, b, c in zip(got, want, again):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
        assert torch.equal(a, c)


def test_roi_align_fpn_gather_backward_full_size():
    """BASELINE shapes (K=1024 cls RoIs 7x7 and 192 grid RoIs 14x14 on the 2 x 256-channel pyramid): gather == scatter,
    and the gradient mass is conserved for in-bounds RoIs."""
    rng = np.random.default_rng(3)
    B, Cc = 2, 256
    sizes = [(200, 336), (100, 168), (50, 84), (25, 42)]
    shapes = [(B, Cc, h, w) for h, w in sizes]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    for K, ph in ((1024, 7), (192, 14)):
        rois = _random_rois(rng, K, B, 1300, 780)
        rois[:, 1:3] = np.abs(rois[:, 1:3]) + 8
        rois[:, 3] = np.minimum(rois[:, 3], 1300)
        rois[:, 4] = np.minimum(rois[:, 4], 780)
        go = torch.randn(K, Cc, ph, ph, device="cuda").contiguous(memory_format=torch.channels_last)
        a = _fpn_backward("gather", go, dev(rois), shapes, scales, ph, 2)
        b = _fpn_backward("atomic", go, dev(rois), shapes, scales, ph, 2)
        for x, y in zip(a, b):
            torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-4)
        total = sum(float(t.double().sum()) for t in a)
        assert abs(total - float(go.double().sum())) < 1e-2 + 1e-4 * float(go.double().abs().sum()) ** 0.5


def test_soft_nms_vs_reference_golden(oracle):
    """Device soft-NMS against the reference's soft_nms.cpp outputs (tests/golden/soft_nms.npz): same survivors in the
    same order; linear / hard scores bit-equal, gaussian within expf rounding."""
    import pet.lib.ops as ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "soft_nms.npz"))
    names = {0: "hard", 1: "linear", 2: "gaussian"}
    for i in range(len([k for k in g.files if k.startswith("c") and k.endswith("_cfg")])):
        method, sigma, thr, ms = g["c%d_cfg" % i]
        b, s, k = ops.soft_nms(dev(g["c%d_boxes" % i]).reshape(-1, 4), dev(g["c%d_scores" % i]), sigma, thr, ms,
                               names[int(method)])
        assert np.array_equal(k.cpu().numpy(), g["c%d_out_idx" % i]), i
        assert np.array_equal(b.cpu().numpy(), g["c%d_out_boxes" % i].reshape(-1, 4)), i
        if int(method) == 2:
            np.testing.assert_allclose(s.cpu().numpy(), g["c%d_out_scores" % i], rtol=2e-6, atol=1e-7)
        else:
            assert np.array_equal(s.cpu().numpy(), g["c%d_out_scores" % i]), i


@pytest.mark.parametrize("method", [0, 1, 2])
def test_soft_nms_segments_vs_oracle(oracle, method):
    """80 class segments of ragged sizes (0 .. 2048 boxes) in two launches, each against the C oracle."""
    import pet.lib.ops as ops
    rng = np.random.default_rng(17 + method)
    sizes = [0, 1, 2, 63, 64, 65, 300, 2048] + list(rng.integers(0, 400, 72))
    boxes = np.concatenate([_rand_boxes(rng, n, 250) for n in sizes]).astype(np.float32)
    scores = rng.uniform(0, 1, boxes.shape[0]).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    for lo in (0, 64):
        seg = offs[lo:lo + 65] if lo + 64 < len(offs) else offs[lo:]
        a, z = int(seg[0]), int(seg[-1])
        b, s, k, c = ops.soft_nms_segments(dev(boxes[a:z]), dev(scores[a:z]), [int(o) - a for o in seg], 0.5, 0.3,
                                           0.02, method)
        c = c.cpu().numpy()
        for j in range(len(seg) - 1):
            o0, o1 = int(seg[j]) - a, int(seg[j + 1]) - a
            wb, wsc, wk = oracle.soft_nms(boxes[a + o0:a + o1], scores[a + o0:a + o1], 0.5, 0.3, 0.02, method)
            assert c[j] == len(wk), (lo, j)
            assert np.array_equal(k[o0:o0 + c[j]].cpu().numpy(), wk)
            assert np.array_equal(b[o0:o0 + c[j]].cpu().numpy(), wb.reshape(-1, 4))
            if method == 2:
                np.testing.assert_allclose(s[o0:o0 + c[j]].cpu().numpy(), wsc, rtol=2e-6, atol=1e-7)
            else:
                assert np.array_equal(s[o0:o0 + c[j]].cpu().numpy(), wsc)


def test_soft_nms_boxlist_and_errors():
    import pet.lib.ops as ops
    from pet.lib.ops.boxlist_ops import boxlist_soft_nms
    from pet.utils.data.structures.bounding_box import BoxList
    rng = np.random.default_rng(5)
    bl = BoxList(dev(_rand_boxes(rng, 50, 100)), (400, 300))
    bl.add_field("scores", dev(rng.uniform(0, 1, 50).astype(np.float32)))
    out = boxlist_soft_nms(bl, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method="linear")
    assert out.has_field("scores") and 0 < len(out) <= 50
    sc = out.get_field("scores")
    assert boxlist_soft_nms(bl, overlap_thresh=0) is bl
    assert float(sc[0]) == float(bl.get_field("scores").max())          # the first pick keeps its score
    e = ops.soft_nms(torch.zeros(0, 4, device="cuda"), torch.zeros(0, device="cuda"))
    assert e[0].shape == (0, 4) and e[2].dtype == torch.int64
    with pytest.raises(RuntimeError):
        ops.soft_nms(torch.zeros(2049, 4, device="cuda"), torch.zeros(2049, device="cuda"))
    with pytest.raises(AssertionError):
        ops.soft_nms(torch.zeros(4, 4, device="cuda"), torch.zeros(4, device="cuda"), method="nope")


@pytest.mark.parametrize("method", ["ID", "TEMP_AVG", "AVG", "IOU_AVG", "GENERALIZED_AVG", "QUASI_SUM"])
def test_box_voting_vs_oracle(oracle, method):
    import pet.lib.ops as ops
    from pet.lib.ops.boxes import BOX_VOTING_METHODS
    rng = np.random.default_rng(23)
    allb = _rand_boxes(rng, 700, 200)
    alls = rng.uniform(0.05, 1, 700).astype(np.float32)
    top = np.sort(rng.choice(700, 90, replace=False))
    beta = 1.5 if method in ("TEMP_AVG", "GENERALIZED_AVG", "QUASI_SUM") else 1.0
    b, s = ops.box_voting(dev(allb[top]), dev(alls[top]), dev(allb), dev(alls), 0.6, method, beta)
    wb, ws = oracle.box_voting(allb[top], alls[top], allb, alls, 0.6, BOX_VOTING_METHODS[method], beta)
    np.testing.assert_allclose(b.cpu().numpy(), wb, rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(s.cpu().numpy(), ws, rtol=2e-5, atol=1e-6)


def test_box_voting_known_answers():
    import pet.lib.ops as ops
    box = torch.tensor([[10., 20., 50., 80.]], device="cuda")
    far = torch.tensor([[300., 300., 340., 350.]], device="cuda")
    b, s = ops.box_voting(box, torch.tensor([0.7], device="cuda"), torch.cat([box, far]),
                          torch.tensor([0.7, 0.9], device="cuda"), 0.5, "AVG")
    assert torch.allclose(b, box) and abs(float(s) - 0.7) < 1e-6              # a lone box votes for itself
    two = torch.tensor([[10., 20., 50., 80.], [12., 22., 52., 82.]], device="cuda")
    b, s = ops.box_voting(two[:1], torch.tensor([0.9], device="cuda"), two, torch.tensor([0.5, 0.5], device="cuda"), 0.5,
                          "ID")
    assert torch.allclose(b, two.mean(0, keepdim=True)) and abs(float(s) - 0.9) < 1e-7   # equal weights: the mean box
    e = ops.box_voting(two[:0], torch.zeros(0, device="cuda"), two, torch.ones(2, device="cuda"), 0.5)
    assert e[0].shape == (0, 4)


@pytest.mark.parametrize("mode", ["ml_nms", "soft", "soft_vote", "vote"])
def test_filter_results_modes(oracle, mode):
    """pet/rcnn/core/test.py filter_results: every branch against a per-class numpy evaluation with the oracle."""
    from pet.rcnn.core import config
    from pet.rcnn.core.test import filter_results
    from pet.utils.data.structures.bounding_box import BoxList
    rng = np.random.default_rng(31)
    C, R = 81, 120
    config.reset_cfg()
    config.merge_cfg_from_list(["MODEL.NUM_CLASSES", C, "FAST_RCNN.SCORE_THRESH", 0.3, "FAST_RCNN.NMS", 0.5,
                                "FAST_RCNN.DETECTIONS_PER_IMG", 60, "TEST.SOFT_NMS.ENABLED", "soft" in mode,
                                "TEST.BBOX_VOTE.ENABLED", "vote" in mode, "TEST.BBOX_VOTE.SCORING_METHOD", "AVG"])
    try:
        base = _rand_boxes(rng, R, 300)
        boxes = np.repeat(base, C, axis=0) + rng.uniform(-2, 2, (R * C, 4)).astype(np.float32)
        scores = (rng.uniform(0, 1, R * C) ** 6).astype(np.float32)
        labels = np.tile(np.arange(C), R)
        bl = BoxList(dev(boxes), (640, 480))
        bl.add_field("scores", dev(scores))
        if mode != "ml_nms":
            bl.add_field("labels", torch.from_numpy(labels).cuda())
        out = filter_results(bl)
        ob, os_, ol = out.bbox.cpu().numpy(), out.get_field("scores").cpu().numpy(), out.get_field("labels").cpu().numpy()
        # numpy evaluation
        wb, ws, wl = [], [], []
        for j in range(1, C):
            m = (labels == j) & (scores > 0.3)
            cb, cs = boxes[m], scores[m]
            if mode in ("ml_nms", "vote"):
                k = oracle.nms(cb, cs, 0.5)
                kb, ks = cb[k], cs[k]
            else:
                kb, ks, _ = oracle.soft_nms(cb, cs, 0.5, 0.5, 0.0001, 1)
            if "vote" in mode and len(cb):
                kb, ks = oracle.box_voting(kb, ks, cb, cs, 0.8, 2, 1.0)
            wb.append(kb.reshape(-1, 4)); ws.append(ks); wl.append(np.full(len(ks), j))
        wb, ws, wl = np.concatenate(wb), np.concatenate(ws), np.concatenate(wl)
        if len(ws) > 60:
            th = np.sort(ws)[len(ws) - 60]
            sel = ws >= th
            wb, ws, wl = wb[sel], ws[sel], wl[sel]
        key_got = np.lexsort((os_, ol))
        key_want = np.lexsort((ws, wl))
        assert len(os_) == len(ws)
        assert np.array_equal(ol[key_got], wl[key_want])
        np.testing.assert_allclose(os_[key_got], ws[key_want], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ob[key_got], wb[key_want], rtol=1e-5, atol=1e-3)
    finally:
        config.reset_cfg()


def test_nms_vs_reference_hard_soft_nms_golden():
    """The device NMS against the reference's own CPU greedy NMS (soft_nms.cpp, 'hard' method; tie-free cases of
    tests/golden/soft_nms.npz): identical keep lists, also through the multi-label entry with a single label."""
    import pet.lib.ops as ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "soft_nms.npz"))
    n_hard = 0
    for i in range(len([k for k in g.files if k.startswith("c") and k.endswith("_cfg")])):
        method, _, thr, min_score = g["c%d_cfg" % i]
        boxes, scores = g["c%d_boxes" % i], g["c%d_scores" % i]
        if int(method) != 0 or len(scores) < 700:
            continue
        n_hard += 1
        live = np.nonzero(scores >= np.float32(min_score))[0]
        b, s = dev(boxes[live]), dev(scores[live])
        assert np.array_equal(live[ops.nms(b, s, float(thr)).cpu().numpy()], g["c%d_out_idx" % i])
        lab = torch.zeros(len(live), dtype=torch.int64, device="cuda")
        assert np.array_equal(live[ops.ml_nms(b, s, lab, float(thr)).cpu().numpy()], g["c%d_out_idx" % i])
    assert n_hard >= 3


def test_ml_soft_nms_vs_reference_golden():
    """Multi-label soft-NMS on the device against the reference's ml_soft_nms.cpp (tests/golden/soft_nms.npz): survivors,
    labels and order identical, scores bit-equal (gaussian: expf rounding), the top-k rule included; and the hard
    method == the device ml_nms keep lists (the reference's CPU form of that op)."""
    import pet.lib.ops as ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "soft_nms.npz"))
    names = {0: "hard", 1: "linear", 2: "gaussian"}
    n_cases = len([k for k in g.files if k.startswith("m") and k.endswith("_cfg")])
    assert n_cases >= 8
    for i in range(n_cases):
        method, sigma, thr, ms, topk = g["m%d_cfg" % i]
        boxes, scores, labels = g["m%d_boxes" % i], g["m%d_scores" % i], g["m%d_labels" % i]
        b, s, l, k = ops.ml_soft_nms(dev(boxes).reshape(-1, 4), dev(scores), torch.from_numpy(labels).cuda(), sigma, thr,
                                     ms, names[int(method)], int(topk))
        assert np.array_equal(k.cpu().numpy(), g["m%d_out_idx" % i]), i
        assert np.array_equal(l.cpu().numpy(), g["m%d_out_labels" % i]), i
        assert np.array_equal(b.cpu().numpy(), g["m%d_out_boxes" % i].reshape(-1, 4)), i
        if int(method) == 2:
            np.testing.assert_allclose(s.cpu().numpy(), g["m%d_out_scores" % i], rtol=2e-6, atol=1e-7)
        else:
            assert np.array_equal(s.cpu().numpy(), g["m%d_out_scores" % i]), i
        if int(method) == 0 and len(scores):
            live = np.nonzero(scores >= np.float32(ms))[0]
            keep = ops.ml_nms(dev(boxes[live]), dev(scores[live]), torch.from_numpy(labels[live]).cuda(), float(thr),
                              max(int(topk), 0))
            assert np.array_equal(live[keep.cpu().numpy()], g["m%d_out_idx" % i]), i


def test_box_ml_voting_equals_per_label_voting(oracle):
    """Multi-label voting == single-label voting run label by label (box_ml_voting.cu:17: other labels have IoU 0)."""
    import pet.lib.ops as ops
    rng = np.random.default_rng(29)
    allb = _rand_boxes(rng, 600, 180)
    alls = rng.uniform(0.05, 1, 600).astype(np.float32)
    alll = rng.integers(1, 5, 600).astype(np.int64)
    top = np.sort(rng.choice(600, 80, replace=False))
    b, s, l = ops.box_ml_voting(dev(allb[top]), dev(alls[top]), torch.from_numpy(alll[top]).cuda(), dev(allb), dev(alls),
                                torch.from_numpy(alll).cuda(), 0.5, "AVG")
    assert np.array_equal(l.cpu().numpy(), alll[top])
    for lab in range(1, 5):
        ti, ai = np.nonzero(alll[top] == lab)[0], np.nonzero(alll == lab)[0]
        wb, ws = oracle.box_voting(allb[top][ti], alls[top][ti], allb[ai], alls[ai], 0.5, 2, 1.0)
        np.testing.assert_allclose(b.cpu().numpy()[ti], wb, rtol=2e-5, atol=1e-4)
        np.testing.assert_allclose(s.cpu().numpy()[ti], ws, rtol=2e-5, atol=1e-6)
