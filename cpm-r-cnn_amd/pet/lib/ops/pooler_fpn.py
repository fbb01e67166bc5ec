"""Fused multi-level RoIAlign: LevelMapper + per-level RoIAlign + scatter in ONE launch.

Replaces the loop in Pooler.forward (pet/rcnn/utils/poolers.py:113-132: 4 `nonzero` syncs, 4 gathers,
4 RoIAlign launches, 4 scatters) with cpm_roi_align_fpn_{forward,backward}.  Feature maps are NHWC.
"""
import ctypes
import os

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


_GROUPED = os.environ.get("CPM_ROI_BWD_GROUP", "1") != "0"
_GATHER_MAX_ROIS, _GATHER_MAX_SETS = 8192, 8      # cpm_roi_align_fpn_backward_gather_sets: per pass, all sets together


class _RoiBackwardGroup(object):
    """The RoIAlign calls of one forward pass on one pyramid whose gradients are formed by ONE pass over the pyramid's
    tiles (cpm_roi_align_fpn_backward_gather_sets): each call's backward node parks its pooled gradient here, the last
    one -- or whoever needs the maps first (H.run_deferred) -- launches.  The gather kernel's time is per tile, not per
    RoI (a workgroup per 8x8 tile walks dependent phases whatever the list length): five calls of a step visit the
    tiles five times, 0.74 ms; together once."""

    def __init__(self):
        self.registered = 0
        self.pending = []
        self.meta = None
        self.holders = None

    def flush(self):
        if not self.pending:
            return
        from . import conv as C
        pend, self.pending = self.pending, []
        scales, lvl_min, lvl_max, (s0, l0, eps), shapes = self.meta
        accs = []
        for h in self.holders:
            C._wait_readers(h)
            accs.append(h["acc"])
        n = len(accs)
        hs, ws, sc = _tables(accs, scales)
        m = len(pend)
        vp, ip = ctypes.c_void_p * m, ctypes.c_int * m
        dev = accs[0].device
        ktot = sum(int(r.shape[0]) for _, r, _, _, _ in pend)
        with H.guard(dev):
            need = H.lib().cpm_roi_align_fpn_gather_workspace_bytes(hs, ws, n, int(shapes[0][0]), ktot)
            wsb = H.workspace(need, dev)
            rc = H.lib().cpm_roi_align_fpn_backward_gather_sets(
                m, vp(*[g.data_ptr() for g, _, _, _, _ in pend]), vp(*[r.data_ptr() for _, r, _, _, _ in pend]),
                ip(*[int(r.shape[0]) for _, r, _, _, _ in pend]), ip(*[p[2] for p in pend]), ip(*[p[3] for p in pend]),
                ip(*[p[4] for p in pend]), (ctypes.c_void_p * n)(*[t.data_ptr() for t in accs]), hs, ws, sc, n,
                int(shapes[0][0]), int(shapes[0][1]), H.f(lvl_min), H.f(lvl_max), H.f(s0), H.f(l0), H.f(eps),
                (1 << n) - 1, H.ptr(wsb), H.c_size_t(wsb.numel()), H.stream())
        H.check(rc, "roi_align_fpn_backward_gather_sets")


def roi_backward_group(feats):
    """Called by the model once per training forward, before its heads pool `feats`: the RoIAlign calls on these maps
    are then differentiated together (see _RoiBackwardGroup).  Only for maps opted into shared gradient accumulation
    (conv.mark_shared_grad): their gradient tensors are written in place, which is what lets a node answer autograd
    before its contribution has been added."""
    if not (_GROUPED and torch.is_grad_enabled() and feats and feats[0].is_cuda):
        return None
    if getattr(feats[0], "_cpm_gacc", None) is None:      # (a call whose maps lack an accumulator goes alone: backward)
        return None
    grp = _RoiBackwardGroup()
    feats[0]._cpm_roi_group = grp
    return grp


class _RoIAlignFPN(Function):
    @staticmethod
    def forward(ctx, rois, output_size, scales, sampling_ratio, lvl_min, lvl_max, canonical, *feats):
        H.require_gpu(rois, *feats)
        ctx.group = getattr(feats[0], "_cpm_roi_group", None) if any(ctx.needs_input_grad[7:]) else None
        if ctx.group is not None:
            ctx.group.registered += 1
        given = feats
        feats = [_nhwc(f) for f in feats]
        B, C = feats[0].shape[:2]
        K = rois.shape[0]
        ph, pw = output_size
        out, levels = H.side_alloc(lambda: (
            torch.empty((K, C, ph, pw), dtype=torch.float32, device=rois.device, memory_format=torch.channels_last),
            torch.empty((max(K, 1),), dtype=torch.int32, device=rois.device)))
        r = rois.contiguous().float()
        if H.in_side_section():
            # (conv.fwd_side) the kernel below is queued on the second stream, the tensors made here are compute-stream
            # blocks: the level indices are dropped by most callers at once, so they are parked until the join; and a
            # layout / dtype copy made just now ran on the COMPUTE stream behind the fork point -- order the second
            # stream behind it
            H.keep(levels, r, *feats)
            if r is not rois or any(a is not b for a, b in zip(feats, given)):
                H.fork(H._raw_stream(rois.device.index), H.stream_raw())
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
            grp = ctx.group
            if grp is not None:                # one call fewer to wait for: the group may be complete now
                grp.registered -= 1
                if grp.registered <= 0:
                    grp.flush()
            return (None,) * (7 + len(shapes))
        g = _nhwc(grad_out)
        K = r.shape[0]
        C = int(shapes[0][1])
        # gather formulation (no float atomics, no zero fill, bit-reproducible) when the shape allows it; the atomic
        # scatter kernel otherwise.  A level whose feature map is opted into shared gradient accumulation and already
        # has an accumulator gets this call's contribution added there (and reports None to autograd).
        gather = ph <= 16 and pw <= 16 and C % 4 == 0 and K <= 8192
        grp = ctx.group
        meta = (scales, lvl_min, lvl_max, (s0, l0, eps), shapes)
        if (grp is not None and gather and all(h is not None for h in ctx.holders)
                and (grp.meta is None or grp.meta == meta)):
            # differentiated together with the group's other calls: make sure every map has its accumulator (the first
            # consumer hands it to autograd, cleared), park the pooled gradient, launch when the group is complete
            rets = []
            for s_, h in zip(shapes, ctx.holders):
                if "acc" in h and tuple(h["acc"].shape) == tuple(s_):
                    rets.append(None)
                else:
                    t = torch.empty(s_, dtype=torch.float32, device=g.device, memory_format=torch.channels_last).zero_()
                    h["acc"] = t
                    rets.append(t)
            if grp.meta is None:
                grp.meta, grp.holders = meta, ctx.holders
            # one pass takes at most _GATHER_MAX_SETS sets and _GATHER_MAX_ROIS RoIs IN ALL (roi_align.hip): a set that
            # would push the parked ones over either limit sends them off first (larger per-GPU batches: 8 images x 512
            # rows for the cls head and as many for the RSM head)
            if grp.pending and (len(grp.pending) >= _GATHER_MAX_SETS
                                or sum(int(p[1].shape[0]) for p in grp.pending) + K > _GATHER_MAX_ROIS):
                grp.flush()
            first = not grp.pending
            grp.pending.append((g, r, int(ph), int(pw), int(ratio)))
            grp.registered -= 1
            if grp.registered <= 0:
                grp.flush()
            elif first:
                H.deferred.append(grp.flush)
                torch.autograd.Variable._execution_engine.queue_callback(H.run_deferred)
            return (None, None, None, None, None, None, None) + tuple(rets)
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
