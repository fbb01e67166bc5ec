"""numpy-facing wrapper around oracle/liboracle.so (cpm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "cpm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_ml_nms.restype = ctypes.c_int64
        _LIB.orc_nms.restype = ctypes.c_int64
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def roi_align_forward(inp, rois, scale, ph, pw, sampling_ratio, aligned=False, interp=0):
    inp, rois = _f32(inp), _f32(rois).reshape(-1, 5)
    B, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.zeros((K, C, ph, pw), np.float32)
    rc = lib().orc_roi_align_forward(_p(inp), _p(rois), K, B, C, H, W, ctypes.c_float(scale), ph, pw,
                                     sampling_ratio, int(aligned), interp, _p(out))
    if rc:
        raise RuntimeError("orc_roi_align_forward failed: %d" % rc)
    return out


def roi_align_backward(grad, rois, scale, ph, pw, B, C, H, W, sampling_ratio, aligned=False, interp=0):
    grad, rois = _f32(grad), _f32(rois).reshape(-1, 5)
    K = rois.shape[0]
    gin = np.zeros((B, C, H, W), np.float32)
    rc = lib().orc_roi_align_backward(_p(grad), _p(rois), K, B, C, H, W, ctypes.c_float(scale), ph, pw,
                                      sampling_ratio, int(aligned), interp, _p(gin))
    if rc:
        raise RuntimeError("orc_roi_align_backward failed: %d" % rc)
    return gin


def ml_nms(boxes, scores, labels, thr, topk=0):
    boxes, scores = _f32(boxes).reshape(-1, 4), _f32(scores)
    labels = np.ascontiguousarray(labels, dtype=np.int64)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    nk = lib().orc_ml_nms(_p(boxes), _p(scores), _p(labels), ctypes.c_int64(n), ctypes.c_float(thr),
                          ctypes.c_int64(topk), _p(keep))
    return keep[:nk].copy()


def nms(boxes, scores, thr):
    boxes, scores = _f32(boxes).reshape(-1, 4), _f32(scores)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    nk = lib().orc_nms(_p(boxes), _p(scores), ctypes.c_int64(n), ctypes.c_float(thr), _p(keep))
    return keep[:nk].copy()


def soft_nms(boxes, scores, sigma=0.5, thr=0.3, min_score=0.001, method=1):
    """-> (boxes [m,4], decayed scores [m], original indices [m]) in the reference's output order."""
    b, sc = _f32(boxes).reshape(-1, 4).copy(), _f32(scores).copy()
    n = b.shape[0]
    idx = np.zeros(max(n, 1), np.int64)
    lib().orc_soft_nms.restype = ctypes.c_int64
    m = lib().orc_soft_nms(_p(b), _p(sc), _p(idx), ctypes.c_int64(n), ctypes.c_float(thr), int(method),
                           ctypes.c_float(sigma), ctypes.c_float(min_score))
    return b[:m].copy(), sc[:m].copy(), idx[:m].copy()


def ml_soft_nms(boxes, scores, labels, sigma=0.5, thr=0.3, min_score=0.001, method=1, topk=-1):
    """-> (boxes [m,4], decayed scores [m], labels [m], original indices [m]) in the reference's output order."""
    b, sc = _f32(boxes).reshape(-1, 4).copy(), _f32(scores).copy()
    lab = np.ascontiguousarray(labels, dtype=np.int64).copy()
    n = b.shape[0]
    idx = np.zeros(max(n, 1), np.int64)
    lib().orc_ml_soft_nms.restype = ctypes.c_int64
    m = lib().orc_ml_soft_nms(_p(b), _p(sc), _p(lab), _p(idx), ctypes.c_int64(n), ctypes.c_float(thr), int(method),
                              ctypes.c_float(sigma), ctypes.c_float(min_score), ctypes.c_int64(topk))
    return b[:m].copy(), sc[:m].copy(), lab[:m].copy(), idx[:m].copy()


def box_iou(boxes, query):
    boxes, query = _f32(boxes).reshape(-1, 4), _f32(query).reshape(-1, 4)
    out = np.zeros((boxes.shape[0], query.shape[0]), np.float32)
    lib().orc_box_iou(_p(boxes), ctypes.c_int64(boxes.shape[0]), _p(query), ctypes.c_int64(query.shape[0]), _p(out))
    return out


def boxlist_iou(b1, b2):
    b1, b2 = _f32(b1).reshape(-1, 4), _f32(b2).reshape(-1, 4)
    out = np.zeros((b1.shape[0], b2.shape[0]), np.float32)
    lib().orc_boxlist_iou(_p(b1), ctypes.c_int64(b1.shape[0]), _p(b2), ctypes.c_int64(b2.shape[0]), _p(out))
    return out


def pool_points_interp_forward(inp, pts, scale):
    inp, pts = _f32(inp), _f32(pts).reshape(-1, 3)
    B, C, H, W = inp.shape
    out = np.zeros((pts.shape[0], C), np.float32)
    lib().orc_pool_points_interp_forward(_p(inp), _p(pts), pts.shape[0], C, H, W, ctypes.c_float(scale), _p(out))
    return out


def pool_points_interp_backward(grad, pts, scale, B, C, H, W):
    grad, pts = _f32(grad), _f32(pts).reshape(-1, 3)
    gin = np.zeros((B, C, H, W), np.float32)
    lib().orc_pool_points_interp_backward(_p(grad), _p(pts), pts.shape[0], B, C, H, W, ctypes.c_float(scale), _p(gin))
    return gin


def level_map(boxes, k_min=2, k_max=5, s0=224, lvl0=4, eps=1e-6):
    boxes = _f32(boxes).reshape(-1, 4)
    out = np.zeros(boxes.shape[0], np.int64)
    lib().orc_level_map(_p(boxes), ctypes.c_int64(boxes.shape[0]), ctypes.c_float(k_min), ctypes.c_float(k_max),
                        ctypes.c_float(s0), ctypes.c_float(lvl0), ctypes.c_float(eps), _p(out))
    return out


def box_decode(codes, boxes, weights=(1., 1., 1., 1.), clip=float(np.log(1000. / 16))):
    codes, boxes = _f32(codes).reshape(-1, 4), _f32(boxes).reshape(-1, 4)
    out = np.zeros_like(codes)
    w = [ctypes.c_float(x) for x in weights]
    lib().orc_box_decode(_p(codes), _p(boxes), ctypes.c_int64(codes.shape[0]), *w, ctypes.c_float(clip), _p(out))
    return out


def box_encode(ref, prop, weights=(1., 1., 1., 1.)):
    ref, prop = _f32(ref).reshape(-1, 4), _f32(prop).reshape(-1, 4)
    out = np.zeros_like(ref)
    w = [ctypes.c_float(x) for x in weights]
    lib().orc_box_encode(_p(ref), _p(prop), ctypes.c_int64(ref.shape[0]), *w, _p(out))
    return out


def matcher(q, high, low, allow_low_quality=False):
    q = _f32(q)
    M, N = q.shape
    out = np.zeros(N, np.int64)
    lib().orc_matcher(_p(q), ctypes.c_int64(M), ctypes.c_int64(N), ctypes.c_float(high), ctypes.c_float(low),
                      int(allow_low_quality), _p(out))
    return out


def sub_regions(grid_points=9, grid_size=3, map_size=56):
    out = np.zeros((grid_points, 4), np.int32)
    lib().orc_sub_regions(grid_points, grid_size, map_size, _p(out))
    return out


def grid_targets(boxes, gt, grid_points=9, map_size=56, radius=1, mapping_ratio=1.0):
    boxes, gt = _f32(boxes).reshape(-1, 4), _f32(gt).reshape(-1, 4)
    R = boxes.shape[0]
    half = map_size // 4 * 2
    out = np.zeros((R, grid_points, half, half), np.float32)
    lib().orc_grid_targets(_p(boxes), _p(gt), ctypes.c_int64(R), grid_points, map_size, radius,
                           ctypes.c_float(mapping_ratio), _p(out))
    return out


def grid_decode(boxes, prob, grid_points=9, map_size=56, mapping_ratio=1.0):
    boxes, prob = _f32(boxes).reshape(-1, 4), _f32(prob)
    R = boxes.shape[0]
    out = np.zeros((R, 4), np.float32)
    lib().orc_grid_decode(_p(boxes), _p(prob), ctypes.c_int64(R), grid_points, map_size,
                          ctypes.c_float(mapping_ratio), _p(out))
    return out


def cell_anchors(stride, sizes, ratios):
    sizes = np.ascontiguousarray(sizes, np.float64)
    ratios = np.ascontiguousarray(ratios, np.float64)
    out = np.zeros((len(ratios) * len(sizes), 4), np.float64)
    lib().orc_cell_anchors(ctypes.c_double(stride), _p(sizes), len(sizes), _p(ratios), len(ratios), _p(out))
    return out.astype(np.float32)


def grid_anchors(grid_hw, stride, cell):
    """anchor_generator.py:73-95: shifts (x,y,x,y) + cell anchors, row-major over (y, x), then A."""
    gh, gw = grid_hw
    sx = np.arange(0, gw * stride, stride, dtype=np.float32)
    sy = np.arange(0, gh * stride, stride, dtype=np.float32)
    yy, xx = np.meshgrid(sy, sx, indexing="ij")
    shifts = np.stack([xx.ravel(), yy.ravel(), xx.ravel(), yy.ravel()], 1)
    return (shifts[:, None, :] + cell[None, :, :]).reshape(-1, 4)


def deform_conv(x, offset, weight, stride=1, pad=1, dil=1, groups=1, deformable_groups=1, dy=None):
    """Deformable conv v1 (NCHW).  Returns y, or (y, dx, doffset, dw) when dy is given.  offset may be None."""
    x, weight = _f32(x), _f32(weight)
    N, C, H, W = x.shape
    K, _, R, S = weight.shape
    P = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Q = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    off = _f32(offset) if offset is not None else None
    y = np.zeros((N, K, P, Q), np.float32)
    null = ctypes.c_void_p(0)
    if dy is None:
        lib().orc_deform_conv(_p(x), _p(off) if off is not None else null, _p(weight), null, N, C, H, W, K, R, S,
                              stride, pad, dil, groups, deformable_groups, 0, _p(y), null, null, null)
        return y
    dy = _f32(dy)
    dx = np.zeros_like(x)
    doff = np.zeros((N, deformable_groups * 2 * R * S, P, Q), np.float32)
    dw = np.zeros_like(weight)
    lib().orc_deform_conv(_p(x), _p(off) if off is not None else null, _p(weight), _p(dy), N, C, H, W, K, R, S,
                          stride, pad, dil, groups, deformable_groups, 1, _p(y), _p(dx), _p(doff), _p(dw))
    return y, dx, doff, dw


# ---- image preparation -------------------------------------------------------------------------------
# The reference resizes with torchvision F.resize -> PIL.Image.resize(BILINEAR) (pet/utils/data/transforms/
# transforms.py:60-64).  Pillow is a third-party dependency that is not vendored in /root/reference; this is a numpy
# restatement of its published algorithm (Pillow src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
# ImagingResampleHorizontal_8bpc / Vertical_8bpc; installed here: Pillow 12.2.0).  PINNED: tests/test_data_pipeline.py
# checks it bit-for-bit against PIL.Image.resize itself (PIL is importable in this image, here and on the GPU box).
_PB = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """Per output position: first tap, tap count, Q22 integer taps (bilinear / triangle filter, support scaled by the
    down-sampling ratio; plain Python loops in C operation order)."""
    import math
    scale = float(in_size) / out_size
    fs = scale if scale >= 1.0 else 1.0
    support = 1.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / fs
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0.0 else t
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << _PB)) if w[x] < 0 else int(0.5 + w[x] * (1 << _PB))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pil_resize_bilinear(img, oh, ow):
    """uint8 [H,W,C] -> uint8 [oh,ow,C]: horizontal pass, then vertical pass on its uint8 result."""
    x = np.asarray(img).astype(np.int64)
    if ow != x.shape[1]:
        b, kk = pil_bilinear_coeffs(x.shape[1], ow)
        out = np.zeros((x.shape[0], ow, x.shape[2]), np.int64)
        for xx in range(ow):
            lo, n = b[xx]
            acc = (1 << (_PB - 1)) + (x[:, lo:lo + n, :] * kk[xx, :n][None, :, None]).sum(1)
            out[:, xx, :] = np.clip(acc >> _PB, 0, 255)
        x = out
    if oh != x.shape[0]:
        b, kk = pil_bilinear_coeffs(x.shape[0], oh)
        out = np.zeros((oh, x.shape[1], x.shape[2]), np.int64)
        for yy in range(oh):
            lo, n = b[yy]
            acc = (1 << (_PB - 1)) + (x[lo:lo + n] * kk[yy, :n][:, None, None]).sum(0)
            out[yy] = np.clip(acc >> _PB, 0, 255)
        x = out
    return x.astype(np.uint8)


def image_prep(img, out_hw, flip, mean, std, to_bgr255, pad_hw):
    """Resize -> hflip -> ToTensor -> Normalize -> zero pad (transforms.py:60-115, image_list.py:56-66) of one uint8 RGB
    image; returns fp32 [3, pad_h, pad_w].  The value arithmetic is fp32: ((v / 255) * 255 - mean) / std."""
    r = pil_resize_bilinear(img, out_hw[0], out_hw[1])
    if flip:
        r = r[:, ::-1, :]
    t = np.ascontiguousarray(r.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
    if to_bgr255:
        t = t[[2, 1, 0]] * np.float32(255)
    t = (t - np.asarray(mean, np.float32).reshape(3, 1, 1)) / np.asarray(std, np.float32).reshape(3, 1, 1)
    out = np.zeros((3, pad_hw[0], pad_hw[1]), np.float32)
    out[:, : t.shape[1], : t.shape[2]] = t
    return out


def box_voting(top_boxes, top_scores, all_boxes, all_scores, thr, method=0, beta=1.0):
    """numpy restatement of pet/lib/ops/csrc/Box_ops/box_voting.cu:24-210 (fp32 per pair, float64 sums).
    PARITY UNPINNED against the reference binary: the op is CUDA-only there and its tests hold no vectors; pinned by
    the known answers in tests/test_gpu_ops.py (a lone box votes for itself; equal weights give the mean box)."""
    tb, ab = _f32(top_boxes).reshape(-1, 4), _f32(all_boxes).reshape(-1, 4)
    ts, as_ = _f32(top_scores), _f32(all_scores)
    iou = box_iou(tb, ab)
    out_b, out_s = np.zeros_like(tb), ts.copy()
    for i in range(tb.shape[0]):
        m = iou[i] >= np.float32(thr)
        w = as_[m]
        sw = np.ones_like(w)
        sc = w.copy()
        if method == 1:
            nz = w != 0
            sc[nz] = np.float32(1) / (np.float32(1) + np.power(np.float32(1) / w[nz] - np.float32(1),
                                                                 np.float32(1.0 / beta), dtype=np.float32))
        elif method == 3:
            sw = iou[i][m]
            sc = iou[i][m] * w
        elif method == 4:
            sc = np.power(w, np.float32(beta), dtype=np.float32)
        bw = w.astype(np.float64).sum()
        out_b[i] = (ab[m].astype(np.float64) * w[:, None].astype(np.float64)).sum(0) / bw
        num, ssum = sw.astype(np.float64).sum(), sc.astype(np.float64).sum()
        if method in (1, 2, 3):
            out_s[i] = ssum / num
        elif method == 4:
            out_s[i] = (ssum / num) ** (1.0 / beta)
        elif method == 5:
            out_s[i] = ssum / num ** beta
    return out_b, out_s


def cv_resize_linear(img, scale, flip=False):
    """numpy float32 restatement of cv2.resize(im.astype(float32), None, None, fx=scale, fy=scale, INTER_LINEAR) as
    used by pet/rcnn/core/test.py:340-358 (optionally on the mirrored image).  OpenCV is a third-party dependency
    absent from /root/reference and from this image: PARITY UNPINNED; the published algorithm (imgproc/resize.cpp:
    dsize = cvRound(size * f); sx = (dx + 0.5) / fx - 0.5; left tap floor(sx), fraction zeroed when the tap is
    clamped at a border; columns blended first, then rows, in float32)."""
    im = np.asarray(img)
    if flip:
        im = im[:, ::-1, :]
    im = im.astype(np.float32)
    H, W = im.shape[:2]
    oh, ow = int(np.rint(H * scale)), int(np.rint(W * scale))
    inv = np.float32(1.0 / scale)

    def taps(n_out, n_in):
        f = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * inv - np.float32(0.5)
        s = np.floor(f).astype(np.int64)
        a = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        s[lo], a[lo] = 0, 0
        hi = s >= n_in - 1
        s[hi], a[hi] = n_in - 1, 0
        return s, np.minimum(s + 1, n_in - 1), a
    x0, x1, ax = taps(ow, W)
    y0, y1, ay = taps(oh, H)
    a0, a1 = (np.float32(1) - ax)[None, :, None], ax[None, :, None]
    top = im[y0][:, x0] * a0 + im[y0][:, x1] * a1
    bot = im[y1][:, x0] * a0 + im[y1][:, x1] * a1
    out = top * (np.float32(1) - ay)[:, None, None] + bot * ay[:, None, None]
    return out.astype(np.float32)


def balanced_sample_quotas(labels, counts, batch_size_per_image, positive_fraction):
    """Per-image sample sizes of BalancedPositiveNegativeSampler
    (pet/rcnn/utils/balanced_positive_negative_sampler.py:36-46): num_pos = min(#(label >= 1), int(batch * fraction)),
    num_neg = min(#(label == 0), batch - num_pos).  The members are a uniformly random subset (torch.randperm,
    :49-50), so the checkable facts are the sizes, the membership (a sampled positive IS a positive) and the
    inclusion frequencies.  labels: 1-d array, image-contiguous; counts: candidates per image.  -> int [images, 2]."""
    labels = np.asarray(labels)
    out, o = [], 0
    for c in counts:
        seg = labels[o:o + c]
        num_pos = min(int((seg >= 1).sum()), int(batch_size_per_image * positive_fraction))
        num_neg = min(int((seg == 0).sum()), batch_size_per_image - num_pos)
        out.append((num_pos, num_neg))
        o += c
    return np.asarray(out, dtype=np.int64).reshape(len(counts), 2)
