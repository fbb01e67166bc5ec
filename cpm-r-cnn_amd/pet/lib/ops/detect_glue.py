"""Fused device kernels for the detection glue of the training step: RoI<->gt matching over all images
of the batch, the grid heat-map loss with on-the-fly targets, and the heat-map -> box decoder.  Host wrappers of
cpm_match_rois / cpm_grid_bce_loss / cpm_grid_decode (include/cpmrcnn_hip.h); the reference's per-image tensor-op
formulations they replace stay available next to them (pet/rcnn/utils/matcher.py,
grid_cascade_rcnn/{loss,inference}.py) and the GPU tests hold the two against each other."""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _hip as H


def _boxes(t):
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != 4:
        raise RuntimeError("boxes must be float32 [R, 4]")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


def match_rois(rois, roi_img, gts, gt_off, high, low, allow_low_quality=False):
    """rois [R,4]; roi_img [R] int32 image index per RoI (None = one image); gts [G,4] concatenated over images with
    gt_off [N+1] int32 (device).  Returns (matched int64 [R]: gt index within the RoI's image / -1 / -2, max IoU)."""
    H.require_gpu(rois, gts)
    rois, gts = _boxes(rois), _boxes(gts)
    R, G = rois.shape[0], gts.shape[0]
    if G == 0:
        raise ValueError("No ground-truth boxes available for one of the images during training")
    matched = torch.empty((R,), dtype=torch.int64, device=rois.device)
    max_iou = torch.empty((R,), dtype=torch.float32, device=rois.device)
    ws = torch.empty((G,), dtype=torch.int32, device=rois.device) if allow_low_quality else None
    roi_img = _i32(roi_img) if roi_img is not None else None
    gt_off = _i32(gt_off)
    with H.guard(rois.device):
        rc = H.lib().cpm_match_rois(H.ptr(rois), H.ptr(roi_img), H.ptr(gts), H.ptr(gt_off), R, G, H.f(high),
                                    H.f(low), int(bool(allow_low_quality)), H.ptr(ws), H.ptr(matched),
                                    H.ptr(max_iou), H.stream())
    H.check(rc, "match_rois")
    return matched, max_iou


def _geom_args(logits, sub_regions):
    if logits.dim() != 4 or logits.dtype != torch.float32:
        raise RuntimeError("grid logits must be float32 [R, P, h, w]")
    strides = (ctypes.c_int64 * 4)(*logits.stride())
    pts = logits.shape[1]
    sub = (ctypes.c_int * (2 * pts))(*[int(v) for s in sub_regions for v in s[:2]])
    return strides, sub, pts


class _GridBCEFn(Function):
    @staticmethod
    def forward(ctx, logits, rois, gt_boxes, map_size, sub_regions, ratio, radius, weight):
        H.require_gpu(logits, rois, gt_boxes)
        rois, gt_boxes = _boxes(rois), _boxes(gt_boxes)
        R = logits.shape[0]
        if rois.shape[0] != R or gt_boxes.shape[0] != R or logits.shape[2] != map_size // 4 * 2:
            raise RuntimeError("grid loss: %d logits maps, %d rois, %d gts" % (R, rois.shape[0], gt_boxes.shape[0]))
        strides, sub, pts = _geom_args(logits, sub_regions)
        loss = torch.zeros((), dtype=torch.float32, device=logits.device)
        grad = torch.empty_like(logits)                      # preserve_format: same strides as the logits
        assert grad.stride() == logits.stride()
        with H.guard(logits.device):
            rc = H.lib().cpm_grid_bce_loss(H.ptr(logits), strides, H.ptr(rois), H.ptr(gt_boxes), R, pts,
                                           int(map_size), sub, H.f(ratio), int(radius), H.f(weight), H.ptr(loss),
                                           H.ptr(grad), H.stream())
        H.check(rc, "grid_bce_loss")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return H.scaled_by(grad, g), None, None, None, None, None, None, None


def grid_bce_loss(logits, rois, gt_boxes, map_size, sub_regions, mapping_ratio, radius, weight=1.0):
    """weight * mean(BCEWithLogits(logits, rasterised point targets)); targets as GridLossComputation.prepare_target."""
    return _GridBCEFn.apply(logits, rois, gt_boxes, map_size, sub_regions, float(mapping_ratio), int(radius),
                            float(weight))


def grid_decode(logits, rois, map_size, sub_regions, mapping_ratio, roi_img=None, gts=None, gt_off=None):
    """Refined boxes [R,4] from the heat maps (GridPostProcessor.get_boxes); with gts also the boolean keep mask of
    GridPostProcessor._filter_boxes evaluated on the incoming RoIs."""
    H.require_gpu(logits, rois, gts)
    rois = _boxes(rois)
    R = rois.shape[0]
    if logits.shape[0] != R:
        raise RuntimeError("grid decode: %d logits maps for %d rois" % (logits.shape[0], R))
    logits = logits.detach()
    strides, sub, pts = _geom_args(logits, sub_regions)
    out = torch.empty((R, 4), dtype=torch.float32, device=rois.device)
    keep = None
    if gts is not None:
        gts = _boxes(gts)
        keep = torch.empty((R,), dtype=torch.uint8, device=rois.device)
        roi_img = _i32(roi_img) if roi_img is not None else None
        gt_off = _i32(gt_off)
    with H.guard(rois.device):
        rc = H.lib().cpm_grid_decode(H.ptr(logits), strides, H.ptr(rois), R, pts, int(map_size), sub,
                                     H.f(mapping_ratio), H.ptr(roi_img), H.ptr(gts), H.ptr(gt_off), H.ptr(out),
                                     H.ptr(keep), H.stream())
    H.check(rc, "grid_decode")
    return (out, keep.bool()) if keep is not None else out


def rpn_decode(reg, topk_idx, anchors, weights, clip, image_sizes):
    """reg [N, A, 4] (contiguous), topk_idx [N, k] int64, anchors [A, 4]; image_sizes list of (w, h).
    Returns decoded + clipped boxes [N, k, 4]."""
    H.require_gpu(reg, anchors)
    N, A = reg.shape[0], reg.shape[1]
    k = topk_idx.shape[1]
    reg = reg if reg.is_contiguous() else reg.contiguous()
    anchors = _boxes(anchors)
    idx = topk_idx.contiguous()
    out = torch.empty((N, k, 4), dtype=torch.float32, device=reg.device)
    w4 = (ctypes.c_float * 4)(*[float(v) for v in weights])
    iw = (ctypes.c_float * N)(*[float(s[0]) for s in image_sizes])
    ih = (ctypes.c_float * N)(*[float(s[1]) for s in image_sizes])
    with H.guard(reg.device):
        rc = H.lib().cpm_rpn_decode(H.ptr(reg), H.ptr(idx), H.ptr(anchors), N, A, k, w4, H.f(clip), iw, ih, H.ptr(out),
                                    H.stream())
    H.check(rc, "rpn_decode")
    return out


def sigmoid_multi(logit_list):
    """sigmoid of several fp32 tensors in one launch (cpm_sigmoid_multi); returns views of one flat buffer, shaped
    like the inputs (which must be dense in memory; the values follow the inputs' memory order)."""
    H.require_gpu(*logit_list)
    L = len(logit_list)
    ns = [t.numel() for t in logit_list]
    offs = [0]
    for n in ns:
        offs.append(offs[-1] + n)
    flat = torch.empty((offs[-1],), dtype=torch.float32, device=logit_list[0].device)
    vp, ip = ctypes.c_void_p * L, ctypes.c_int * L
    with H.guard(flat.device):
        rc = H.lib().cpm_sigmoid_multi(vp(*[t.data_ptr() for t in logit_list]), ip(*ns), ip(*offs[:-1]), L,
                                       H.ptr(flat), H.stream())
    H.check(rc, "sigmoid_multi")
    return [flat[offs[i]:offs[i + 1]] for i in range(L)]


def rpn_decode_multi(regs, idxs, anchors, out_offs, out_boxes, weights, clip, image_sizes):
    """rpn_decode for all levels in one launch: regs[l] [N, A_l, 4], idxs[l] [N, k_l] int64, anchors[l] [A_l, 4];
    level l's boxes are written to out_boxes[out_offs[l] : out_offs[l] + N * k_l]."""
    L, N = len(regs), regs[0].shape[0]
    vp, ip = ctypes.c_void_p * L, ctypes.c_int * L
    w4 = (ctypes.c_float * 4)(*[float(v) for v in weights])
    iw = (ctypes.c_float * N)(*[float(s[0]) for s in image_sizes])
    ih = (ctypes.c_float * N)(*[float(s[1]) for s in image_sizes])
    for r, i in zip(regs, idxs):
        if not (r.is_contiguous() and i.is_contiguous()):
            raise RuntimeError("rpn_decode_multi: contiguous inputs")
    with H.guard(out_boxes.device):
        rc = H.lib().cpm_rpn_decode_multi(vp(*[r.data_ptr() for r in regs]), vp(*[i.data_ptr() for i in idxs]),
                                          vp(*[a.data_ptr() for a in anchors]), ip(*[r.shape[1] for r in regs]),
                                          ip(*[i.shape[1] for i in idxs]), ip(*[int(o) for o in out_offs]), L, N, w4,
                                          H.f(clip), iw, ih, H.ptr(out_boxes), H.stream())
    H.check(rc, "rpn_decode_multi")
    return out_boxes


def rpn_labels(matched, visible, discard_between=True):
    """anchor labels (1 / 0 / -1, fp32) from the match and the visibility mask (cpm_rpn_labels)"""
    lab = torch.empty(matched.shape, dtype=torch.float32, device=matched.device)
    vis = None
    if visible is not None:
        vis = visible.contiguous().view(torch.uint8)
    with H.guard(matched.device):
        rc = H.lib().cpm_rpn_labels(H.ptr(matched), H.ptr(vis), ctypes.c_int64(matched.numel()),
                                    int(bool(discard_between)), H.ptr(lab), H.stream())
    H.check(rc, "rpn_labels")
    return lab


TOPK_MAX = 2048


def topk_rows(scores, k):
    """scores [rows, n] fp32 -> (values [rows, k] descending, indices [rows, k] int64), ties by ascending index.
    One launch for all rows (cpm_topk_rows); replaces objectness.topk(k, dim=1, sorted=True) of the RPN
    (pet/rcnn/modeling/rpn/inference.py:79-84).  k <= 2048."""
    H.require_gpu(scores)
    if scores.dim() != 2 or scores.dtype != torch.float32:
        raise RuntimeError("topk_rows: scores must be fp32 [rows, n]")
    rows, n = scores.shape
    if not 1 <= k <= min(n, TOPK_MAX):
        raise RuntimeError("topk_rows: k must be in [1, min(n, %d)], got %d (n = %d)" % (TOPK_MAX, k, n))
    s = scores if scores.is_contiguous() else scores.contiguous()
    vals = torch.empty((rows, k), dtype=torch.float32, device=s.device)
    idx = torch.empty((rows, k), dtype=torch.int64, device=s.device)
    with H.guard(s.device):
        rc = H.lib().cpm_topk_rows(H.ptr(s), rows, n, k, H.ptr(vals), H.ptr(idx), H.stream())
    H.check(rc, "topk_rows")
    return vals, idx


def _topk_ws_bytes(levels, rows):
    f = H.lib().cpm_topk_rows_multi_workspace_bytes
    f.restype = ctypes.c_size_t
    return int(f(int(levels), int(rows)))


def topk_rows_multi(score_list, ks, out=None):
    """topk_rows for several [rows, n_l] matrices (same rows) in one launch (cpm_topk_rows_multi): the RPN's FPN levels.
    Returns a list of (values, indices); `out` may supply that list ([rows, k_l] fp32 / int64, contiguous)."""
    if not score_list or len(score_list) != len(ks) or len(ks) > 8:
        raise RuntimeError("topk_rows_multi: 1..8 matrices with one k each")
    H.require_gpu(*score_list)
    rows = score_list[0].shape[0]
    ss, outs = [], []
    for s, k in zip(score_list, ks):
        if s.dim() != 2 or s.dtype != torch.float32 or s.shape[0] != rows:
            raise RuntimeError("topk_rows_multi: scores must be fp32 [rows, n] with the same rows")
        if not 1 <= k <= min(s.shape[1], TOPK_MAX):
            raise RuntimeError("topk_rows_multi: k must be in [1, min(n, %d)], got %d (n = %d)" % (TOPK_MAX, k, s.shape[1]))
        s = s if s.is_contiguous() else s.contiguous()
        ss.append(s)
        if out is not None:
            v, i = out[len(outs)]
            if (v.shape != (rows, k) or i.shape != (rows, k) or v.dtype != torch.float32 or i.dtype != torch.int64
                    or not v.is_contiguous() or not i.is_contiguous()):
                raise RuntimeError("topk_rows_multi: out must hold contiguous [rows, k] fp32 / int64 pairs")
            outs.append((v, i))
            continue
        outs.append((torch.empty((rows, k), dtype=torch.float32, device=s.device),
                     torch.empty((rows, k), dtype=torch.int64, device=s.device)))
    L = len(ss)
    vp = ctypes.c_void_p * L
    ip = ctypes.c_int * L
    with H.guard(ss[0].device):
        # (long rows -- the finest FPN level -- are selected by several workgroups each: scratch for their candidates)
        need = _topk_ws_bytes(L, rows) if max(s.shape[1] for s in ss) >= 32768 else 0
        wsb = H.workspace(need, ss[0].device) if need else None
        rc = H.lib().cpm_topk_rows_multi(vp(*[s.data_ptr() for s in ss]), ip(*[s.shape[1] for s in ss]),
                                         ip(*[int(k) for k in ks]), L, rows, vp(*[o[0].data_ptr() for o in outs]),
                                         vp(*[o[1].data_ptr() for o in outs]), H.ptr(wsb) if need else None,
                                         H.c_size_t(wsb.numel() if need else 0), H.stream())
    H.check(rc, "topk_rows_multi")
    return outs


class _RPNLossFn(torch.autograd.Function):
    """(sum of BCE terms, sum of smooth-L1 terms) over the sampled anchors, gradients produced by the same launch."""

    @staticmethod
    def forward(ctx, logits, reg, anchors, matched, gts, gt_off, pos, neg, per_image, weights, beta, quota):
        H.require_gpu(logits, reg, anchors, gts)
        total = logits.numel()
        logits_c = logits.contiguous()
        reg_c = reg.contiguous()
        sums = torch.empty(2, dtype=torch.float32, device=logits.device)
        dlog = torch.empty(total, dtype=torch.float32, device=logits.device)
        dreg = torch.empty((total, 4), dtype=torch.float32, device=logits.device)
        w4 = (ctypes.c_float * 4)(*[float(v) for v in weights])
        with H.guard(logits.device):
            rc = H.lib().cpm_rpn_loss(H.ptr(logits_c), H.ptr(reg_c), H.ptr(_boxes(anchors)), H.ptr(matched.contiguous()),
                                      H.ptr(_boxes(gts)), H.ptr(gt_off.contiguous()), H.ptr(pos.contiguous()),
                                      H.ptr(neg.contiguous()), H.c_int64(total), int(per_image), w4, H.f(beta),
                                      H.ptr(sums), H.ptr(dlog), H.ptr(dreg),
                                      H.ptr(quota) if quota is not None else None,
                                      quota.numel() if quota is not None else 0, H.stream())
        H.check(rc, "rpn_loss")
        ctx.save_for_backward(dlog, dreg)
        ctx.shapes = (logits.shape, reg.shape)
        return sums[0], sums[1]

    @staticmethod
    def backward(ctx, g_obj, g_box):
        dlog, dreg = ctx.saved_tensors
        ls, rs = ctx.shapes
        return (H.scaled_by(dlog, g_obj).view(ls), H.scaled_by(dreg, g_box).view(rs), None, None, None, None, None, None,
                None, None, None, None)


def rpn_loss(logits, reg, anchors, matched, gts, gt_off, pos, neg, per_image, weights, beta, quota=None):
    """logits [T], reg [T,4], anchors [T,4], matched int64 [T], gts [G,4], gt_off int32 [images+1], pos / neg bool [T].
    Returns the two sums (BCE over pos|neg, smooth-L1 over pos), differentiable w.r.t. logits and reg: UNDIVIDED without
    `quota`; with the sampler's quota ([images, 2] int32, device) divided by its total inside the kernel -- the two losses
    of loss.py:121-126 as they are."""
    if matched.dtype != torch.int64 or pos.dtype != torch.bool or neg.dtype != torch.bool or gt_off.dtype != torch.int32:
        raise RuntimeError("rpn_loss: matched int64, gt_off int32, pos / neg bool")
    if quota is not None:
        if quota.dtype != torch.int32 or not quota.is_contiguous() or quota.device != logits.device:
            raise RuntimeError("rpn_loss: quota int32, contiguous, on the logits' device")
    return _RPNLossFn.apply(logits, reg, anchors, matched, gts, gt_off, pos, neg, per_image, weights, beta, quota)


_SAMPLE_WS = {}
_LABEL_DTYPES = {torch.float32: 0, torch.int64: 1, torch.int32: 2}


def sample_pos_neg(labels, counts, batch_size_per_image, positive_fraction, seed=None, cand_target=0):
    """labels [R] (float32 / int64 / int32: >= 1 positive, 0 negative, < 0 ignored), image-contiguous with `counts`
    elements per image (host ints).  Returns (pos [R] bool, neg [R] bool, quota [images, 2] int32 = (n_pos, n_neg)):
    BalancedPositiveNegativeSampler (pet/rcnn/utils/balanced_positive_negative_sampler.py:27-67) for all images in
    three launches and no host round trip (cpm_sample_pos_neg).  seed: None draws one from torch's CPU generator
    (torch.manual_seed makes the run reproducible); the same seed gives the same sample."""
    H.require_gpu(labels)
    if labels.dim() != 1 or labels.dtype not in _LABEL_DTYPES:
        raise RuntimeError("sample_pos_neg: labels must be 1-d float32 / int64 / int32")
    counts = [int(c) for c in counts]
    n_img = len(counts)
    if sum(counts) != labels.numel() or n_img < 1:
        raise RuntimeError("sample_pos_neg: counts must sum to len(labels)")
    offs = (ctypes.c_int64 * (n_img + 1))()
    for i, c in enumerate(counts):
        offs[i + 1] = offs[i] + c
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # CPU generator: no device traffic
    dev = labels.device
    lab = labels if labels.is_contiguous() else labels.contiguous()
    pos = torch.empty(lab.numel(), dtype=torch.bool, device=dev)
    neg = torch.empty(lab.numel(), dtype=torch.bool, device=dev)
    quota = torch.empty((n_img, 2), dtype=torch.int32, device=dev)
    ws = _SAMPLE_WS.get(dev)
    if ws is None:
        ws = _SAMPLE_WS[dev] = torch.empty(int(H.lib().cpm_sample_pos_neg_workspace_bytes()), dtype=torch.uint8,
                                           device=dev)
    with H.guard(dev):
        rc = H.lib().cpm_sample_pos_neg(H.ptr(lab), _LABEL_DTYPES[lab.dtype], offs, n_img, int(batch_size_per_image),
                                        int(batch_size_per_image * positive_fraction), ctypes.c_uint64(seed),
                                        int(cand_target), H.ptr(pos), H.ptr(neg), H.ptr(quota), H.ptr(ws), H.stream())
    H.check(rc, "sample_pos_neg")
    return pos, neg, quota
