"""Fused multi-level RoIAlign: LevelMapper + per-level RoIAlign + scatter in ONE launch.

Replaces the loop in Pooler.forward (pet/rcnn/utils/poolers.py:113-132: 4 `nonzero` syncs, 4 gathers,
4 RoIAlign launches, 4 scatters) with cpm_roi_align_fpn_{forward,backward}.  Feature maps are NHWC.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _hip as H


def _tables(feats, scales):
    n = len(feats)
    hs = (ctypes.c_int * n)(*[int(f.shape[2]) for f in feats])
    ws = (ctypes.c_int * n)(*[int(f.shape[3]) for f in feats])
    sc = (ctypes.c_float * n)(*[float(s) for s in scales])
    return hs, ws, sc


def _nhwc(t):
    return t if (t.is_contiguous(memory_format=torch.channels_last)) else t.contiguous(
        memory_format=torch.channels_last)


class _RoIAlignFPN(Function):
    @staticmethod
    def forward(ctx, rois, output_size, scales, sampling_ratio, lvl_min, lvl_max, canonical, *feats):
        H.require_gpu(rois, *feats)
        feats = [_nhwc(f) for f in feats]
        B, C = feats[0].shape[:2]
        K = rois.shape[0]
        ph, pw = output_size
        out = torch.empty((K, C, ph, pw), dtype=torch.float32, device=rois.device,
                          memory_format=torch.channels_last)
        levels = torch.empty((max(K, 1),), dtype=torch.int32, device=rois.device)
        r = rois.contiguous().float()
        n = len(feats)
        hs, ws, sc = _tables(feats, scales)
        ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in feats])
        s0, l0, eps = canonical
        with H.guard(rois.device):
            rc = H.lib().cpm_roi_align_fpn_forward(ptrs, hs, ws, sc, n, H.ptr(r), K, int(B), int(C), ph, pw,
                                                   int(sampling_ratio), H.f(lvl_min), H.f(lvl_max), H.f(s0),
                                                   H.f(l0), H.f(eps), H.ptr(out), H.ptr(levels), H.stream())
        H.check(rc, "roi_align_fpn_forward")
        ctx.holders = [getattr(f, "_cpm_gacc", None) for f in feats]      # see conv.mark_shared_grad
        ctx.save_for_backward(r)
        ctx.meta = (output_size, tuple(float(s) for s in scales), int(sampling_ratio), lvl_min, lvl_max, canonical,
                    [tuple(f.shape) for f in feats])
        lv = levels[:K]
        ctx.mark_non_differentiable(lv)
        ctx.set_materialize_grads(False)       # (no zero-filled "gradient" of the level indices per backward call)
        return out, lv

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, _):
        r, = ctx.saved_tensors
        (ph, pw), scales, ratio, lvl_min, lvl_max, (s0, l0, eps), shapes = ctx.meta
        if grad_out is None:                   # nobody differentiated the pooled features
            return (None,) * (7 + len(shapes))
        g = _nhwc(grad_out)
        K = r.shape[0]
        C = int(shapes[0][1])
        # gather formulation (no float atomics, no zero fill, bit-reproducible) when the shape allows it; the atomic
        # scatter kernel otherwise.  A level whose feature map is opted into shared gradient accumulation and already
        # has an accumulator gets this call's contribution added there (and reports None to autograd).
        gather = ph <= 16 and pw <= 16 and C % 4 == 0 and K <= 8192
        grads, fresh = [], []
        for s, h in zip(shapes, ctx.holders):
            if h is not None and "acc" in h and tuple(h["acc"].shape) == tuple(s):
                grads.append(h["acc"])
                fresh.append(False)
            else:
                t = torch.empty(s, dtype=torch.float32, device=g.device, memory_format=torch.channels_last)
                if not gather:
                    t.zero_()
                if h is not None:
                    h["acc"] = t
                grads.append(t)
                fresh.append(True)
        n = len(grads)
        hs, ws, sc = _tables(grads, scales)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in grads])
        with H.guard(g.device):
            if gather:
                need = H.lib().cpm_roi_align_fpn_gather_workspace_bytes(hs, ws, n, int(shapes[0][0]), K)
                wsb = H.workspace(need, g.device)
                mask = sum((0 if f else 1) << i for i, f in enumerate(fresh))
                rc = H.lib().cpm_roi_align_fpn_backward_gather(
                    H.ptr(g), ptrs, hs, ws, sc, n, H.ptr(r), K, int(shapes[0][0]), C, ph, pw, ratio, H.f(lvl_min),
                    H.f(lvl_max), H.f(s0), H.f(l0), H.f(eps), mask, H.ptr(wsb), H.c_size_t(wsb.numel()), H.stream())
            else:
                rc = H.lib().cpm_roi_align_fpn_backward(H.ptr(g), ptrs, hs, ws, sc, n, H.ptr(r), K,
                                                        int(shapes[0][0]), C, ph, pw, ratio, H.f(lvl_min),
                                                        H.f(lvl_max), H.f(s0), H.f(l0), H.f(eps), H.stream())
        H.check(rc, "roi_align_fpn_backward")
        return (None, None, None, None, None, None, None) + tuple(t if f else None for t, f in zip(grads, fresh))


_LEVEL_RANGE = {}


def _level_range(s_first, s_last):
    """-log2 of the first / last scale in fp32 (poolers.py:84-88), computed once per pair of scales: the two tensor
    ops cost ~30 us of host time per call, right behind each of the step's host round trips"""
    key = (s_first, s_last)
    r = _LEVEL_RANGE.get(key)
    if r is None:
        r = _LEVEL_RANGE[key] = (-float(torch.log2(torch.tensor(s_first, dtype=torch.float32))),
                                 -float(torch.log2(torch.tensor(s_last, dtype=torch.float32))))
    return r


def roi_align_fpn(feats, rois, output_size, scales, sampling_ratio, canonical_scale=224, canonical_level=4,
                  eps=1e-6, return_levels=False):
    """feats: list of [B,C,H_l,W_l]; rois [K,5].  Level range follows poolers.py:84-88 (-log2 of the scales)."""
    lvl_min, lvl_max = _level_range(float(scales[0]), float(scales[len(feats) - 1]))
    out, levels = _RoIAlignFPN.apply(rois, tuple(output_size), tuple(scales), sampling_ratio, lvl_min, lvl_max,
                                     (float(canonical_scale), float(canonical_level), float(eps)), *feats)
    return (out, levels) if return_levels else out
