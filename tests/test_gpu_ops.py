"""GPU parity tests (run with -m gpu on the MI355X box): HIP RoIAlign / NMS / box ops through the
C ABI (pet.lib.ops._C -> libcpmrcnn_hip.so) against the CPU oracle and the reference goldens."""
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
