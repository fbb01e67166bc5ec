"""Pins the CPU oracle (oracle/cpm_oracle.c) to vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU-only."""
import os

import numpy as np
import pytest


@pytest.mark.parametrize("ph", [7, 14])
@pytest.mark.parametrize("inv_scale", [4, 8, 16, 32])
def test_roi_align_matches_reference(golden_ops, oracle, ph, inv_scale):
    g = golden_ops
    key = "roi_%d_%d" % (ph, inv_scale)
    x, rois = g[key + "_x"], g[key + "_rois"]
    B, C, H, W = x.shape
    for interp in (0, 1):
        for aligned in (0, 1):
            sfx = "_i%d_a%d" % (interp, aligned)
            y = oracle.roi_align_forward(x, rois, 1.0 / inv_scale, ph, ph, 2, bool(aligned), interp)
            assert np.array_equal(y, g[key + sfx + "_y"])           # bit-exact
            gi = oracle.roi_align_backward(g[key + sfx + "_g"], rois, 1.0 / inv_scale, ph, ph, B, C, H, W, 2,
                                           bool(aligned), interp)
            assert np.array_equal(gi, g[key + sfx + "_gi"])


def test_roi_align_adaptive_sampling(golden_ops, oracle):
    g = golden_ops
    y = oracle.roi_align_forward(g["roi_adapt_x"], g["roi_adapt_rois"], 0.25, 7, 7, 0, False, 0)
    assert np.array_equal(y, g["roi_adapt_y"])


def test_level_mapper(golden_ops, oracle):
    assert np.array_equal(oracle.level_map(golden_ops["lvl_boxes"]), golden_ops["lvl_out"])


def test_anchors(golden_ops, oracle):
    g = golden_ops
    table = np.array([[-83, -39, 100, 56], [-175, -87, 192, 104], [-359, -183, 376, 200], [-55, -55, 72, 72],
                      [-119, -119, 136, 136], [-247, -247, 264, 264], [-35, -79, 52, 96], [-79, -167, 96, 184],
                      [-167, -343, 184, 360]], np.float32)      # anchor_generator.py:210-218 (matlab, 1-based)
    table = table - 1        # the code works 0-based: anchor_generator.py:239 subtracts 1
    assert np.array_equal(g["anchors_matlab_table"], table)
    assert np.array_equal(oracle.cell_anchors(16, (128, 256, 512), (0.5, 1, 2)), table)
    grids = [(5, 7), (3, 4), (2, 2), (1, 2), (1, 1)]
    for i, (size, stride) in enumerate(zip((32, 64, 128, 256, 512), (4, 8, 16, 32, 64))):
        cell = oracle.cell_anchors(stride, (size,), (0.5, 1.0, 2.0))
        assert np.array_equal(cell, g["cell_anchors_%d" % i])
        assert np.array_equal(oracle.grid_anchors(grids[i], stride, cell), g["grid_anchors_%d" % i])


def test_box_coder(golden_ops, oracle):
    g = golden_ops
    dec = oracle.box_decode(g["bc_codes"], g["bc_boxes"])
    np.testing.assert_allclose(dec, g["bc_decode"], rtol=2e-6, atol=1e-4)   # libm expf vs torch's vectorised exp
    enc = oracle.box_encode(g["bc_gt"], g["bc_boxes"])
    np.testing.assert_allclose(enc, g["bc_encode"], rtol=2e-6, atol=1e-6)


def test_iou_and_matcher(golden_ops, oracle):
    g = golden_ops
    iou = oracle.boxlist_iou(g["iou_gt"], g["iou_props"])
    assert np.array_equal(iou, g["iou_out"])
    assert np.array_equal(oracle.matcher(iou, 0.7, 0.3, True), g["match_rpn"])
    assert np.array_equal(oracle.matcher(iou, 0.5, 0.5, False), g["match_cls"])
    assert np.array_equal(oracle.matcher(iou, 0.7, 0.7, False), g["match_g2"])


def test_sub_regions(golden_ops, oracle):
    assert np.array_equal(oracle.sub_regions(9, 3, 56), golden_ops["sub_regions"])


@pytest.mark.parametrize("stage,ratio", [(0, 1.0), (1, 0.5), (2, 0.25)])
def test_grid_targets_and_decode(golden_ops, oracle, stage, ratio):
    g = golden_ops
    t = oracle.grid_targets(g["grid_boxes"], g["grid_gt"], 9, 56, 1, ratio)
    assert np.array_equal(t, g["grid_targets_s%d" % stage])
    assert t[0].sum() == 0 and t[3].sum() > 0
    logits = g["grid_logits_s%d" % stage].astype(np.float64)
    prob = (1.0 / (1.0 + np.exp(-logits))).astype(np.float32)
    dec = oracle.grid_decode(g["grid_boxes"], prob, 9, 56, ratio)
    np.testing.assert_allclose(dec, g["grid_decode_s%d" % stage], rtol=1e-5, atol=1e-3)


def test_nms_against_bruteforce(oracle):
    """nms / ml_nms are parity-unpinned (no CPU kernel in the reference): check the restatement
    against an independent O(N^2) python greedy following ml_nms.cu:11-26,127-140."""
    rng = np.random.default_rng(3)
    for n in (1, 5, 64, 65, 300):
        xy = rng.uniform(0, 200, (n, 2))
        wh = rng.uniform(5, 80, (n, 2))
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        scores = rng.uniform(0, 1, n).astype(np.float32)
        scores[: n // 4] = scores[0]          # ties
        labels = rng.integers(1, 4, n).astype(np.int64)
        order = sorted(range(n), key=lambda i: (-scores[i], i))

        def iou(a, b):
            l, r = max(a[0], b[0]), min(a[2], b[2])
            t, bt = max(a[1], b[1]), min(a[3], b[3])
            w, h = max(np.float32(r - l), np.float32(0)), max(np.float32(bt - t), np.float32(0))
            inter = np.float32(w * h)
            sa = np.float32((a[2] - a[0]) * (a[3] - a[1]))
            sb = np.float32((b[2] - b[0]) * (b[3] - b[1]))
            return np.float32(inter / np.float32(np.float32(sa + sb) - inter))
        for use_labels in (False, True):
            keep, dead = [], set()
            for ii, i in enumerate(order):
                if i in dead:
                    continue
                keep.append(i)
                for j in order[ii + 1:]:
                    if j in dead or (use_labels and labels[i] != labels[j]):
                        continue
                    if iou(boxes[i], boxes[j]) > np.float32(0.5):
                        dead.add(j)
            got = oracle.ml_nms(boxes, scores, labels, 0.5) if use_labels else oracle.nms(boxes, scores, 0.5)
            assert got.tolist() == keep
        assert oracle.ml_nms(boxes, scores, labels, 0.5, topk=3).tolist() == oracle.ml_nms(boxes, scores, labels, 0.5)[:3].tolist()
    assert oracle.nms(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).size == 0


def test_soft_nms_oracle_is_reference(oracle):
    """orc_soft_nms against the reference's own soft_nms.cpp (compiled into oracle/_ref by build_ref.py; vectors in
    tests/golden/soft_nms.npz): surviving boxes, decayed scores and original indices, in output order, bit for bit."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "soft_nms.npz"))
    n_cases = len([k for k in g.files if k.startswith("c") and k.endswith("_cfg")])
    assert n_cases >= 9
    for i in range(n_cases):
        method, sigma, thr, ms = g["c%d_cfg" % i]
        b, s, k = oracle.soft_nms(g["c%d_boxes" % i], g["c%d_scores" % i], sigma, thr, ms, int(method))
        assert np.array_equal(k, g["c%d_out_idx" % i]), i
        assert np.array_equal(b, g["c%d_out_boxes" % i].reshape(-1, 4)), i
        assert np.array_equal(s, g["c%d_out_scores" % i]), i


def test_nms_oracle_is_reference_hard_soft_nms(oracle):
    """Greedy NMS pinned against the reference's own CPU kernel: soft_nms.cpp with the 'hard' method zeroes every box
    whose IoU with the current best exceeds the threshold and drops it -- plain greedy NMS with first-position ties --
    so its surviving indices (selection order = descending score) are what orc_nms / orc_ml_nms must return."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "soft_nms.npz"))
    hard = [i for i in range(len([k for k in g.files if k.startswith("c") and k.endswith("_cfg")]))
            if int(g["c%d_cfg" % i][0]) == 0 and len(g["c%d_scores" % i]) >= 700]       # the tie-free hard cases
    assert len(hard) >= 3
    for i in hard:
        _, _, thr, min_score = g["c%d_cfg" % i]
        boxes, scores = g["c%d_boxes" % i], g["c%d_scores" % i]
        live = np.nonzero(scores >= np.float32(min_score))[0]      # soft_nms.cpp also drops scores under min_score
        assert np.array_equal(live[oracle.nms(boxes[live], scores[live], thr)], g["c%d_out_idx" % i]), i
        lab = np.zeros(len(live), np.int64)
        assert np.array_equal(live[oracle.ml_nms(boxes[live], scores[live], lab, thr)], g["c%d_out_idx" % i]), i


def test_ml_soft_nms_oracle_is_reference(oracle):
    """orc_ml_soft_nms against the reference's ml_soft_nms.cpp (oracle/_ref): survivors, scores, labels, indices in
    output order, bit for bit -- including its top-k rule (`topk == i`: 0 keeps nothing, < 0 never stops).  Its hard
    method is the reference's own CPU form of the multi-label NMS, which pins orc_ml_nms with real labels."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "soft_nms.npz"))
    n_cases = len([k for k in g.files if k.startswith("m") and k.endswith("_cfg")])
    assert n_cases >= 8
    for i in range(n_cases):
        method, sigma, thr, ms, topk = g["m%d_cfg" % i]
        boxes, scores, labels = g["m%d_boxes" % i], g["m%d_scores" % i], g["m%d_labels" % i]
        b, s, l, k = oracle.ml_soft_nms(boxes, scores, labels, sigma, thr, ms, int(method), int(topk))
        assert np.array_equal(k, g["m%d_out_idx" % i]), i
        assert np.array_equal(l, g["m%d_out_labels" % i]), i
        assert np.array_equal(b, g["m%d_out_boxes" % i].reshape(-1, 4)), i
        assert np.array_equal(s, g["m%d_out_scores" % i]), i
        if int(method) == 0 and len(scores):
            live = np.nonzero(scores >= np.float32(ms))[0]
            keep = live[oracle.ml_nms(boxes[live], scores[live], labels[live], thr, 0)]
            want = g["m%d_out_idx" % i]
            if int(topk) > 0:
                keep = keep[: int(topk)]
            assert np.array_equal(keep, want), i
