"""Convolution / linear / transposed-conv / GroupNorm operators over the implicit-GEMM HIP kernels.

These replace the ATen (cuDNN/cuBLAS in the reference) calls behind nn.Conv2d, nn.Linear,
nn.ConvTranspose2d and nn.GroupNorm on the hot path.  Activations are
NHWC in memory (torch channels_last, logical shape stays [N,C,H,W]); weights are KRSC in memory
(torch [K,C/g,R,S] channels_last).  Everything is fp32; autograd is wired with explicit backward
kernels (data gradient, weight gradient, fused-epilogue gradient).
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _hip as H

CL = torch.channels_last


def mark_shared_grad(t):
    """Opt a multi-consumer activation into in-place gradient accumulation: the first of this package's consumers to
    run its backward hands autograd its gradient tensor and parks it in the holder; every later consumer adds into
    that same tensor (residual epilogue of the data-gradient kernel / RoIAlign atomics) and reports None, instead of
    returning a fresh tensor for autograd to sum with an extra elementwise pass.  Only for tensors whose consumers
    created before any foreign (non-accumulating) consumer are all from this package -- block inputs of the
    ResNet/ResNeXt bottlenecks and the FPN maps (their one foreign consumer, the P6 subsample, is created first and
    therefore differentiated last)."""
    if torch.is_grad_enabled() and t.requires_grad:
        t._cpm_gacc = {}
    return t


def nhwc(t):
    """Return `t` with NHWC memory (no copy when it already is)."""
    if t.dim() != 4:
        raise RuntimeError("expected a 4-D tensor")
    return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)


def empty_nhwc(shape, like):
    return H.side_alloc(lambda: torch.empty(shape, dtype=torch.float32, device=like.device, memory_format=CL))


def out_size(h, k, stride, pad, dil=1):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


def make_desc(n, c, h, w, k, r, s, stride, pad, dil, groups):
    d = H.ConvDesc()
    d.N, d.H, d.W, d.C, d.K, d.R, d.S = int(n), int(h), int(w), int(c), int(k), int(r), int(s)
    d.stride, d.pad, d.dilation, d.groups = int(stride), int(pad), int(dil), int(groups)
    d.P, d.Q = out_size(h, r, stride, pad, dil), out_size(w, s, stride, pad, dil)
    return d


def _ws(desc, device):
    nbytes = H.lib().cpm_conv2d_workspace_bytes(H.ctypes.byref(desc))
    if nbytes == 0:
        raise RuntimeError("invalid convolution descriptor: %s" % H.lib().cpm_last_error().decode())
    return H.workspace(nbytes, device)


def split_w4(w, out=None):
    """The pre-split (bf16 hi / lo) image of a dense float tensor (cpm_split_w4): same shape / strides / bytes."""
    out = H.side_alloc(lambda: torch.empty_like(w)) if out is None else out
    assert w.numel() % 4 == 0 and out.numel() == w.numel()
    with H.guard(w.device):
        rc = H.lib().cpm_split_w4(H.ptr(w), H.ptr(out), H.ctypes.c_int64(w.numel()), H.stream())
    H.check(rc, "split_w4")
    return out


_W4 = os.environ.get("CPM_W4", "1") != "0"


bf16x3 = H.bf16x3


def w4_of(w_in, w, cg):
    """The pre-split image of a conv / Linear weight for the bf16x3 kernels, or None (other arithmetic, channels per
    group % 4 != 0, a repacked copy of the parameter).  Cached on the parameter with its `_version`: parameters of the
    flat optimizer get theirs from the SGD kernel itself (FlatSGD.step: a view of its image buffer, refreshed with
    every update); anything else -- frozen weights, weights before the first optimizer step, modules outside a
    trainer -- is split here on first use and again whenever the tensor was modified in place."""
    if not _W4 or w is not w_in or cg % 4 or not bf16x3():
        return None
    own = getattr(w_in, "_cpm_owner", w_in)          # a Linear's per-call [K,C,1,1] view stands for its parameter
    t = getattr(own, "_cpm_w4", None)
    if t is not None and own._cpm_w4_version == own._version and own._cpm_w4_ptr == own.data_ptr():
        return t
    src = own.detach()
    if t is None or t.numel() != src.numel() or t.device != src.device:
        t = H.side_alloc(lambda: torch.empty(src.numel(), dtype=torch.float32, device=src.device))
    split_w4(src if src.dim() != 4 else _wmem(src), out=t)
    own._cpm_w4, own._cpm_w4_version, own._cpm_w4_ptr = t, own._version, own.data_ptr()
    return t


def conv2d_forward(x, w, scale, shift, residual, res_mode, relu, stride, pad, dil, groups, w4=None):
    """y = epilogue(conv(x, w)).  w4: the pre-split image of w (split_w4; bf16x3 arithmetic only), used in place of w."""
    n, c, h, wd = x.shape
    k, _, r, s = w.shape
    d = make_desc(n, c, h, wd, k, r, s, stride, pad, dil, groups)
    y = empty_nhwc((n, k, d.P, d.Q), x)
    if y.numel() == 0:
        return y
    ws = _ws(d, x.device)
    with H.guard(x.device):
        if w4 is not None:
            rc = H.lib().cpm_conv2d_forward_w4(H.ctypes.byref(d), H.ptr(x), H.ptr(w4), H.ptr(scale), H.ptr(shift),
                                               H.ptr(residual), int(res_mode), int(bool(relu)), H.ptr(y), H.ptr(ws),
                                               H.c_size_t(ws.numel()), H.stream())
        else:
            rc = H.lib().cpm_conv2d_forward(H.ctypes.byref(d), H.ptr(x), H.ptr(w), H.ptr(scale), H.ptr(shift),
                                            H.ptr(residual), int(res_mode), int(bool(relu)), H.ptr(y), H.ptr(ws),
                                            H.c_size_t(ws.numel()), H.stream())
    H.check(rc, "conv2d_forward")
    return y


_WT_CACHE = os.environ.get("CPM_DGRAD_WT_CACHE", "1") != "0"
# CPM_GATE_BY_CONSUMERS=0: multi-consumer ReLU outputs (bottleneck outputs, the RPN's shared conv) run their own
# epilogue-backward pass again (A/B switch; the sole-consumer chains are not affected)
_GATE_BY_CONSUMERS = os.environ.get("CPM_GATE_BY_CONSUMERS", "1") != "0"


_pending_wt_events = {}         # device index -> events of weight-image transforms nobody has waited for yet


def set_pending_wt_event(ev, device=None):
    """The optimizer's once-per-step weight-image transform runs on the second stream (FlatSGD._refresh_dgrad_weights):
    whoever reads an image on that DEVICE next makes its stream wait for this event first (wait_pending_wt).  Kept per
    device, and as a list: a process may drive several devices, and several optimizers (bench.py's other-body legs)
    may have transforms outstanding on one."""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    _pending_wt_events.setdefault(idx, []).append(ev)


def wait_pending_wt(device=None):
    """make the current stream of `device` wait for every weight-image transform queued on its second stream"""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    evs = _pending_wt_events.pop(idx, None)
    if evs:
        st = torch.cuda.current_stream(idx)
        for ev in evs:
            st.wait_event(ev)


# Every backward route that writes into a flat optimizer's gradient buffer starts with a forward call that counts a use
# of the parameter (_note_use).  FlatSGD.zero_grad skips its memset only when nothing was counted since the SGD kernel
# cleared the buffer (FlatSGD.clear_grads_in_step): a backward pass run outside a trainer step -- the warm-up
# iterations of a hipGraph capture, a gradient probe -- makes the next zero_grad clear for real.
_grad_write_generation = 0


def grad_write_generation():
    return _grad_write_generation


def _note_use(p):
    global _grad_write_generation
    _grad_write_generation += 1
    p._cpm_uses = getattr(p, "_cpm_uses", 0) + 1


def _prepared_call(wparam):
    """the data-gradient entry point that reads the prepared image of `wparam` (float or pre-split, FlatSGD decides)"""
    own = getattr(wparam, "_cpm_owner", wparam)
    return (H.lib().cpm_conv2d_backward_data_prepared_w4 if getattr(own, "_cpm_wt_fmt", 0)
            else H.lib().cpm_conv2d_backward_data_prepared)


def _prepared_wt(wparam, groups, kg, rs, cg, k_scale=None):
    """The data-gradient image of a weight owned by the flat optimizer, made for ALL such weights in one launch after
    every optimizer step (pet/utils/optimizer.py: FlatSGD._refresh_dgrad_weights) -- or None (transform per call):
    first use (this call registers the weight with its convolution geometry), foreign weights, or a weight modified
    since the last refresh (`_version` moved: load_state_dict, manual edits).  k_scale: the frozen per-output-channel
    factor behind the conv; the image is that of diag(k_scale) * W (cpm_conv2d_backward_data_fused)."""
    if wparam is None or not _WT_CACHE:
        return None
    wparam = getattr(wparam, "_cpm_owner", wparam)      # a Linear's per-call [K,C,1,1] view stands for its parameter
    key = (groups, kg, rs, cg, 0 if k_scale is None else k_scale.data_ptr())
    reg = getattr(wparam, "_cpm_wt_desc", None)
    if reg is None:
        if getattr(wparam, "_cpm_grad_sink", None) is not None:
            wparam._cpm_wt_desc = key
            wparam._cpm_wt_scale = k_scale
        return None
    wt = getattr(wparam, "_cpm_wt", None)
    if wt is None or reg != key or wparam._cpm_wt_version != wparam._version:
        return None
    if k_scale is not None and getattr(wparam, "_cpm_wt_scale_version", None) != k_scale._version:
        return None
    if getattr(wparam, "_cpm_wt_fmt", 0) and not bf16x3():      # a pre-split image serves the bf16x3 kernels only
        return None
    wait_pending_wt(wt.device)                         # first reader after an optimizer step: the transform must be done
    return wt


def conv2d_backward_data(dy, w, x_shape, stride, pad, dil, groups, accumulate_into=None, wparam=None, k_scale=None,
                         gate=None):
    """dx = conv^T(k_scale * dy, w); with `accumulate_into` (an NHWC tensor of x's shape) the result is ADDED to it
    instead, and with `gate` (the conv's input x = relu(..)) the running sum is masked: acc = (acc + dx) * [gate > 0].
    `wparam`: the parameter `w` is (a view of), for the once-per-step weight image (_prepared_wt).
    k_scale [K]: the frozen per-channel factor behind the conv (dy is the gradient at conv * k_scale + shift)."""
    n, c, h, wd = x_shape
    k, _, r, s = w.shape
    full_window = (r, s) == (h, wd) and pad == 0 and stride == 1 and dil == 1 and groups == 1 and (r > 1 or s > 1)
    wt = None if full_window else _prepared_wt(wparam, groups, k // groups, r * s, c // groups, k_scale)
    if accumulate_into is not None:
        acc = accumulate_into
        assert tuple(acc.shape) == tuple(x_shape) and acc.is_contiguous(memory_format=CL)
        if dy.numel() == 0 or acc.numel() == 0:
            return acc
        d = make_desc(n, c, h, wd, k, r, s, stride, pad, dil, groups)
        with H.guard(dy.device):
            ws = _ws(d, dy.device)
            if wt is not None:
                rc = _prepared_call(wparam)(H.ctypes.byref(d), H.ptr(dy), H.ptr(wt), H.ptr(acc), 1,
                                            None, H.ptr(gate), H.ptr(ws), H.c_size_t(ws.numel()), H.stream())
            else:
                rc = H.lib().cpm_conv2d_backward_data_fused(H.ctypes.byref(d), H.ptr(dy), H.ptr(w), H.ptr(k_scale),
                                                            H.ptr(acc), 1, None, H.ptr(gate), H.ptr(ws),
                                                            H.c_size_t(ws.numel()), H.stream())
        H.check(rc, "conv2d_backward_data(accumulate)")
        return acc
    assert gate is None, "a gated data gradient without accumulation: conv2d_backward_data_gated"
    if full_window:
        # full-window conv (an FC over a flattened NHWC map): every input pixel sees exactly one tap, so the data
        # gradient is the plain GEMM dy[N,K] x W[K, R*S*C] -- run it as a 1x1 problem over R*S*C "channels"
        # (the KRSC weight bytes are already that matrix) instead of 49 taps of which 48 are masked per row
        w2 = w.permute(0, 2, 3, 1).reshape(k, r * s * c, 1, 1)
        dx2 = conv2d_backward_data(dy.reshape(n, k, 1, 1), w2, (n, r * s * c, 1, 1), 1, 0, 1, 1, wparam=wparam,
                                   k_scale=k_scale)
        return dx2.view(n, r, s, c).permute(0, 3, 1, 2)
    d = make_desc(n, c, h, wd, k, r, s, stride, pad, dil, groups)
    dx = empty_nhwc((n, c, h, wd), dy)
    if dx.numel() == 0:
        return dx
    if dy.numel() == 0:
        return dx.zero_()
    with H.guard(dy.device):
        ws = _ws(d, dy.device)
        if wt is not None:
            rc = _prepared_call(wparam)(H.ctypes.byref(d), H.ptr(dy), H.ptr(wt), H.ptr(dx), 0, None,
                                        None, H.ptr(ws), H.c_size_t(ws.numel()), H.stream())
        elif k_scale is not None:
            rc = H.lib().cpm_conv2d_backward_data_fused(H.ctypes.byref(d), H.ptr(dy), H.ptr(w), H.ptr(k_scale),
                                                        H.ptr(dx), 0, None, None, H.ptr(ws), H.c_size_t(ws.numel()),
                                                        H.stream())
        else:
            rc = H.lib().cpm_conv2d_backward_data(H.ctypes.byref(d), H.ptr(dy), H.ptr(w), H.ptr(dx), 0, H.ptr(ws),
                                                  H.c_size_t(ws.numel()), H.stream())
    H.check(rc, "conv2d_backward_data")
    return dx


def conv2d_backward_data_gated(dy, w, x, in_scale, stride, pad, dil, groups, wparam=None, k_scale=None):
    """Data gradient with the producer's ReLU gate (x > 0) (and an optional factor on dx) folded into the epilogue;
    k_scale as in conv2d_backward_data."""
    n, c, h, wd = x.shape
    k, _, r, s = w.shape
    wt = _prepared_wt(wparam, groups, k // groups, r * s, c // groups, k_scale)
    d = make_desc(n, c, h, wd, k, r, s, stride, pad, dil, groups)
    dx = empty_nhwc((n, c, h, wd), dy)
    if dx.numel() == 0:
        return dx
    if dy.numel() == 0:
        return dx.zero_()
    with H.guard(dy.device):
        ws = _ws(d, dy.device)
        if wt is not None:
            rc = _prepared_call(wparam)(H.ctypes.byref(d), H.ptr(dy), H.ptr(wt), H.ptr(dx), 0,
                                        H.ptr(in_scale), H.ptr(x), H.ptr(ws), H.c_size_t(ws.numel()), H.stream())
        else:
            rc = H.lib().cpm_conv2d_backward_data_fused(H.ctypes.byref(d), H.ptr(dy), H.ptr(w), H.ptr(k_scale),
                                                        H.ptr(dx), 0, H.ptr(in_scale), H.ptr(x), H.ptr(ws),
                                                        H.c_size_t(ws.numel()), H.stream())
    H.check(rc, "conv2d_backward_data_gated")
    return dx


def conv2d_backward_weight(x, dy, w_like, stride, pad, dil, groups, out=None, dbias=None, k_scale=None):
    """dw (+)= k_scale * (x (*) dy).  `out` (same strides as the weight) is accumulated into when given.
    dbias [K]: the bias gradient sum_m dy[m, k] is ADDED to it by the same launch (cpm_conv2d_backward_weight_bias).
    k_scale [K]: the frozen per-channel factor behind the conv (dy is the gradient at conv * k_scale + shift)."""
    n, c, h, wd = x.shape
    k, _, r, s = w_like.shape
    d = make_desc(n, c, h, wd, k, r, s, stride, pad, dil, groups)
    dw = out if out is not None else torch.zeros_like(w_like)
    if x.numel() == 0 or dy.numel() == 0:
        return dw
    ws = _ws(d, x.device)
    with H.guard(x.device):
        if k_scale is not None:
            rc = H.lib().cpm_conv2d_backward_weight_scaled(H.ctypes.byref(d), H.ptr(x), H.ptr(dy), H.ptr(k_scale),
                                                           H.ptr(dw), H.ptr(dbias), H.ptr(ws), H.c_size_t(ws.numel()),
                                                           H.stream())
        elif dbias is not None:
            rc = H.lib().cpm_conv2d_backward_weight_bias(H.ctypes.byref(d), H.ptr(x), H.ptr(dy), H.ptr(dw),
                                                         H.ptr(dbias), H.ptr(ws), H.c_size_t(ws.numel()), H.stream())
        else:
            rc = H.lib().cpm_conv2d_backward_weight(H.ctypes.byref(d), H.ptr(x), H.ptr(dy), H.ptr(dw), H.ptr(ws),
                                                    H.c_size_t(ws.numel()), H.stream())
    H.check(rc, "conv2d_backward_weight")
    return dw


def epilogue_backward(dy, y, scale, relu, want_dpre=True, want_dres=False, want_dshift=False, dshift_out=None):
    """Backward of relu?(v*scale+shift+res): returns (dpre, dres, dshift).  `dshift_out` ([K], accumulated into)
    replaces the fresh zero-filled bias-gradient buffer."""
    k = dy.shape[1]
    m = dy.numel() // max(k, 1)
    dpre = torch.empty_like(dy) if want_dpre else None
    dres = torch.empty_like(dy) if want_dres else None
    dshift = None
    if want_dshift:
        dshift = dshift_out if dshift_out is not None else torch.zeros((k,), dtype=torch.float32, device=dy.device)
    if dy.numel():
        with H.guard(dy.device):
            rc = H.lib().cpm_epilogue_backward(H.ptr(dy), H.ptr(y), H.ptr(scale), int(bool(relu)), H.c_int64(m), k,
                                               H.ptr(dpre), H.ptr(dres), H.ptr(dshift), H.stream())
        H.check(rc, "epilogue_backward")
    return dpre, dres, dshift


def _sink_of(p, needed):
    """The parameter's slice of the flat gradient buffer, when the flat optimizer owns it and a gradient is needed."""
    if not needed or p is None:
        return None
    s = getattr(p, "_cpm_grad_sink", None)
    if s is None or s.data_ptr() == 0 or not s.is_contiguous():
        return None
    _note_use(p)
    return p


def _sink_done(p):
    p._cpm_uses -= 1
    if p._cpm_uses == 0:
        ready = getattr(p, "_cpm_grad_ready", None)
        if ready is not None:
            ready(p)


def _wmem(w):
    """weights must be KRSC in memory"""
    return w if w.is_contiguous(memory_format=CL) else w.contiguous(memory_format=CL)


# ---- weight gradients beside the data-gradient chain --------------------------------------------------------------
# The backward pass is one dependent chain of data gradients; a layer's weight gradient hangs off it as a leaf that
# nothing waits for until the optimizer step.  Most layers of this model fill less than the 256 CUs on their own
# (M = 2..8 k pixels in layer3 / layer4 / the RoI heads), so the weight-gradient kernels go to a second stream and
# run in the gaps: forked after the layer's data gradient is queued, joined once at the end of the backward pass.
_SIDE_WGRAD = os.environ.get("CPM_WGRAD_STREAM", "1") != "0"
_side = {}              # device index -> (torch stream, raw handle)
_side_armed = {}        # device index -> raw compute stream that has to wait for the side work of the running backward


def wgrad_stream(device):
    """the second stream weight gradients run on (None when disabled): whoever consumes gradients on a stream other
    than the compute stream (pet.utils.parallel) must wait for it too"""
    if not _SIDE_WGRAD:
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _side.get(idx)
    if st is None:
        # CPM_WGRAD_PRIO: queue priority of the second stream (lower number = higher priority; out-of-range values map to
        # the nearest valid one): the weight gradients are leaves nothing waits for until the optimizer step, the
        # data-gradient chain on the compute stream is the critical path
        # CPM_WGRAD_RESERVE_CUS=n: the stream's kernels stay off n compute units (cpm_stream_create_cu_reserve), which
        # a data-parallel run leaves to the RCCL kernels reducing finished gradient chunks beside the backward pass
        reserve = int(os.environ.get("CPM_WGRAD_RESERVE_CUS", "0"))
        if reserve > 0:
            out = H.c_void_p()
            with H.guard(device):
                H.check(H.lib().cpm_stream_create_cu_reserve(reserve, H.ctypes.byref(out)), "stream_create_cu_reserve")
            t = torch.cuda.ExternalStream(out.value, device=device)
        else:
            t = torch.cuda.Stream(device=device, priority=int(os.environ.get("CPM_WGRAD_PRIO", "0")))
        st = _side[idx] = (t, t.cuda_stream)
        H.register_aux_stream(device, t)
    return st[0]


# ---- a second queue in the FORWARD pass ---------------------------------------------------------------------------
# Independent branches of the forward pass -- a stage's downsample conv beside conv1 -> conv2, an FPN level's 3x3 output
# conv beside the lateral chain below it, the RPN's shared conv on P3..P6 beside P2's -- each fill a fraction of the chip
# (30-60 us kernels on 30-130 workgroups).  `fwd_fork(t)` marks the point in the compute stream a branch may start behind
# (everything queued so far: its inputs); the ops inside `with fwd_side(t):` are then queued on the package's second
# stream (the one the weight gradients use in backward, idle in forward); `fwd_join(t)` makes the compute stream wait
# for them before their consumer is queued.  Only the KERNELS move: torch's current stream is untouched, so the autograd
# nodes belong to the compute stream and the backward pass is what it was (in-place gradient accumulators, gradient
# sinks and the weight-gradient fork / join assume one compute stream there).
# THE RULE for code that can run inside a side section (three live bugs came from breaking it):
#  * it creates tensors only through H.side_alloc(): the block then comes from the SECOND stream's allocator pool and
#    has the compute stream recorded on it.  A plain torch.empty would ask the compute stream's pool, which may hand out
#    a block that compute-stream kernels queued behind the fork point are still reading;
#  * a tensor it creates and does NOT return or save (RoIAlign's level indices, a layout copy) is parked with H.keep()
#    until the join;
#  * a copy of an input made on the compute stream inside a section (a weight that is not KRSC in memory) goes through
#    _side_copies(): the side stream is ordered behind the copy and the copy is parked.
# Results are bit-identical with the switch on or off.
_FWD_SIDE = os.environ.get("CPM_FWD_SIDE", "1") != "0"


_FWD_SIDE_MASK = int(os.environ.get("CPM_FWD_SIDE_MASK", "31"))     # debugging: which callers may fork (1 body downsample,
#                                                                      2 FPN output convs, 4 RPN levels, 8 cls head, 16 RSM head)


def fwd_fork(t, who=31):
    """the second stream waits for everything queued on the compute stream so far; False when the switch is off"""
    if not (_FWD_SIDE and _SIDE_WGRAD and t.is_cuda and (_FWD_SIDE_MASK & who)):
        return False
    idx = t.device.index
    st = _side.get(idx)
    if st is None:
        wgrad_stream(t.device)
        st = _side[idx]
    H.fork(H._raw_stream(idx), st[1])
    return True


def fwd_side(t):
    """`with fwd_side(t):` -- ops inside are queued on the second stream (call fwd_fork first, fwd_join afterwards);
    the tensors they create come from that stream's allocator pool (H.side_alloc)"""
    st = _side[t.device.index]
    return H.use_stream(st[1], alloc_on=st[0])


def fwd_join(t):
    """the compute stream waits for the second stream"""
    idx = t.device.index
    H.fork(_side[idx][1], H._raw_stream(idx))
    H.release_kept()            # temporaries of the side sections: compute-stream reuse is ordered behind the join now


def _join_side():
    """end of the backward pass: the compute stream waits for the weight gradients"""
    for idx, main_raw in list(_side_armed.items()):
        H.fork(_side[idx][1], main_raw)
    _side_armed.clear()
    H.release_retired()         # side-stream workspaces replaced during this pass: the join above orders their reuse
    H.release_kept()


def _wgrad_on_side(x, dy, w, stride, pad, dil, groups, out, dbias, k_scale=None, want_event=False):
    """The weight gradient on the second stream.  want_event: an event behind it on that stream -- `dy` is about to
    be parked as a shared gradient accumulator that later consumers write IN PLACE on the compute stream; they wait
    for this event first (_wait_readers)."""
    dev = x.device
    idx = dev.index
    st = _side.get(idx)
    if st is None:
        wgrad_stream(dev)
        st = _side[idx]
    main_raw = H._raw_stream(idx)                   # the stream this op's forward ran on (autograd made it current)
    H.fork(main_raw, st[1])                         # everything queued so far: dy, the gate pass
    with H.use_stream(st[1]):
        conv2d_backward_weight(x, dy, w, stride, pad, dil, groups, out=out, dbias=dbias, k_scale=k_scale)
    ev = None
    if want_event:
        ev = torch.cuda.Event()
        ev.record(st[0])
    # the caching allocator must not hand these blocks to the compute stream while the side stream still reads them
    x.record_stream(st[0])
    dy.record_stream(st[0])
    # one join per fork is queued (the first one to run does the work, the rest find nothing armed): a backward pass
    # that died half-way must not leave a stale "already armed" mark behind that would skip the join of the next one
    _side_armed[idx] = main_raw
    torch.autograd.Variable._execution_engine.queue_callback(_join_side)
    return ev


def _wait_readers(h):
    """before an in-place update of the shared accumulator h["acc"] on the compute stream: wait for the side-stream
    weight gradient that still reads it (the tensor was that layer's own incoming gradient)"""
    ev = h.pop("rd_event", None)
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


def _side_copies(*pairs):
    """(given, used) tensor pairs of an op that may run inside a forward side section (fwd_side): a layout / contiguity
    copy made just now ran on the COMPUTE stream behind the fork point and is a compute-stream block to the allocator --
    order the second stream behind it and park it until the join (H.keep); no-op outside a section or without copies"""
    if not H.in_side_section():
        return
    made = [u for g, u in pairs if u is not None and u is not g]
    if made:
        H.keep(*made)
        idx = made[0].device.index
        H.fork(H._raw_stream(idx if idx is not None else torch.cuda.current_device()), H.stream_raw())


class _ConvFn(Function):
    """y = relu?( conv(x, w) * scale + shift + residual )   (scale/shift/residual optional).
    scale is a frozen per-channel factor (AffineChannel2d) and never receives a gradient; shift receives one
    only when it is a trainable bias."""

    @staticmethod
    def forward(ctx, x, w, scale, shift, residual, stride, pad, dil, groups, relu, res_mode, out_tag=None):
        H.require_gpu(x, w, scale, shift, residual)
        x_in = x
        x = nhwc(x)
        w_in = w
        w = _wmem(w)
        res = nhwc(residual) if residual is not None else None
        _side_copies((x_in, x), (w_in, w), (residual, res))
        # a parameter owned by the flat optimizer carries `_cpm_grad_sink` (its slice of the flat gradient buffer):
        # the weight-gradient kernel then accumulates straight into it (no temporary, no autograd add) and the
        # data-parallel reducer is told when the last use of the step has been accumulated
        ctx.wparam = w_in if (ctx.needs_input_grad[1] and getattr(w_in, "_cpm_grad_sink", None) is not None) else None
        if ctx.wparam is not None:
            own = getattr(w_in, "_cpm_owner", w_in)     # a Linear's [K,C,1,1] view counts on the parameter itself
            _note_use(own)
        # the parameter itself (not a repacked copy) when `w` aliases it: key of the once-per-step dgrad image
        ctx.wsrc = w_in if (w is w_in and getattr(w_in, "_cpm_grad_sink", None) is not None) else None
        ctx.bparam = _sink_of(shift, shift is not None and ctx.needs_input_grad[3])
        y = conv2d_forward(x, w, scale, shift, res, res_mode, relu, stride, pad, dil, groups,
                           w4=w4_of(w_in, w, x.shape[1] // groups))
        # Epilogue-backward folded into the consumers.  y = relu(..) may carry a tag (conv2d: sole_consumer /
        # gate_by_consumers): every consuming _ConvFn then masks its data gradient with (y > 0) in the kernel's epilogue
        # -- the running sum of the shared accumulator when it adds to one: masking is linear and idempotent -- and
        # records in the tag whether everything accumulated so far is masked.  This layer's backward then takes dy as
        # the gradient at its pre-activation without a pass of its own.  Its own frozen scale needs no pass either: it
        # is folded into the reductions (k_scale of the data- and weight-gradient kernels).
        ctx.in_tag = getattr(x_in, "_cpm_epi", None)
        ctx.out_tag = out_tag                       # the same dict is attached to y by conv2d() after apply()
        ctx.x_holder = getattr(x_in, "_cpm_gacc", None)
        ctx.res_holder = getattr(residual, "_cpm_gacc", None) if residual is not None else None
        ctx.res_tag = getattr(residual, "_cpm_epi", None) if residual is not None else None
        ctx.cfg = (stride, pad, dil, groups, relu, res_mode, tuple(x.shape),
                   None if residual is None else tuple(residual.shape))
        ctx.has = (scale is not None, shift is not None, residual is not None)
        ctx.save_for_backward(x, w, scale, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        if H.deferred:                      # gradient work parked by earlier nodes (pooler_fpn) may be part of dy
            H.run_deferred()
        x, w, scale, y = ctx.saved_tensors
        stride, pad, dil, groups, relu, res_mode, x_shape, res_shape = ctx.cfg
        has_scale, has_shift, has_res = ctx.has
        need_x, need_w, _, need_shift, need_res = ctx.needs_input_grad[:5]
        dy = nhwc(dy)
        # g = dy * [y > 0]: the gradient at the pre-activation sum v*scale + shift + res;  dres = g;  dshift = sum g;
        # the conv itself sees scale * g, which the kernels form inside their reductions (k_scale)
        k_scale = scale if has_scale else None
        need_gate = relu and not (ctx.out_tag is not None and ctx.out_tag["applied"])
        want_shift = has_shift and need_shift
        want_res = has_res and need_res
        g, dshift = dy, None
        # parameters whose gradient this call completes: announced (_sink_done) at the END, when the data gradient -- the
        # last reader of the weight -- has been queued too: a listener may update the parameter right away
        # (pet.utils.parallel: the chunk's all-reduce and SGD step on a side stream)
        done = []
        # a bias gradient alone (no gate to apply) rides on the weight-gradient kernel's dy reads
        fuse_bias = (want_shift and not need_gate and need_w and groups == 1 and x_shape[1] > 1 and x.numel() > 0
                     and dy.numel() > 0)
        if need_gate or (want_shift and not fuse_bias):
            bp = ctx.bparam if want_shift else None
            g_k, _, dshift = epilogue_backward(dy, y, None, need_gate, want_dpre=need_gate, want_dres=False,
                                               want_dshift=want_shift,
                                               dshift_out=bp._cpm_grad_sink if bp is not None else None)
            if bp is not None:
                dshift = None                   # accumulated in place
                done.append(bp)
            if need_gate:
                g = g_k
        # the weight gradient first: forked onto the second stream it starts together with the data gradient below
        dw = None
        rd_event = None
        if need_w:
            wp = ctx.wparam
            dbias = None
            if fuse_bias:
                bp = ctx.bparam
                dbias = bp._cpm_grad_sink if bp is not None else torch.zeros(w.shape[0], dtype=torch.float32,
                                                                             device=dy.device)
                if bp is None:
                    dshift = dbias
            if wp is not None and wp._cpm_grad_sink.data_ptr() != 0 and w.data_ptr() == wp.data_ptr():
                if _SIDE_WGRAD and x.numel() and g.numel() and not (fuse_bias and ctx.bparam is None):
                    # g itself becomes the residual branch's gradient below (no copy): whoever later updates it in
                    # place on the compute stream must wait for this read
                    rd_event = _wgrad_on_side(x, g, w, stride, pad, dil, groups, wp._cpm_grad_sink, dbias, k_scale,
                                              want_event=want_res and res_mode == 0)
                else:
                    conv2d_backward_weight(x, g, w, stride, pad, dil, groups, out=wp._cpm_grad_sink, dbias=dbias,
                                           k_scale=k_scale)
                done.append(getattr(wp, "_cpm_owner", wp))
            else:
                dw = conv2d_backward_weight(x, g, w, stride, pad, dil, groups, dbias=dbias, k_scale=k_scale)
                if wp is not None:                      # this use reaches the parameter through autograd's accumulation
                    own = getattr(wp, "_cpm_owner", wp)
                    own._cpm_uses -= 1
            if fuse_bias and ctx.bparam is not None:
                done.append(ctx.bparam)                 # accumulated in place
        gres = None
        if want_res:
            gres = g if res_mode == 0 else upsample2x_add_backward(g, res_shape)
            if ctx.res_tag is not None:
                ctx.res_tag["applied"] = False          # an unmasked contribution to the residual tensor's gradient
            h = ctx.res_holder
            if h is not None:
                if "acc" in h:
                    _wait_readers(h)
                    h["acc"].add_(gres)
                    gres = None
                else:
                    h["acc"] = gres
                    if rd_event is not None:
                        h["rd_event"] = rd_event
        dx = None
        if need_x:
            h = ctx.x_holder
            tag = ctx.in_tag
            full_window = (w.shape[2], w.shape[3]) == (x_shape[2], x_shape[3]) and (w.shape[2] > 1 or w.shape[3] > 1)
            can_gate = tag is not None and dil == 1 and not full_window
            # a producer outside _ConvFn (pet.lib.ops.deform_conv) may still ask its one consumer for its frozen scale
            # too: tag["scale"], applied with the gate -- only possible without accumulation
            legacy_scale = tag.get("scale") if tag is not None else None
            if h is not None and "acc" in h and tuple(h["acc"].shape) == tuple(x_shape):
                can_gate = can_gate and legacy_scale is None
                _wait_readers(h)
                conv2d_backward_data(g, w, x_shape, stride, pad, dil, groups, accumulate_into=h["acc"],
                                     wparam=ctx.wsrc, k_scale=k_scale, gate=x if can_gate else None)
                if tag is not None:
                    if not can_gate:
                        tag["applied"] = False
                    elif stride == 1:
                        tag["applied"] = True           # the whole running sum is masked now
                    # stride > 1: masked where this conv reaches; what the tag said about the rest still holds
            else:
                if can_gate:
                    dx = conv2d_backward_data_gated(g, w, x, legacy_scale, stride, pad, dil, groups, wparam=ctx.wsrc,
                                                    k_scale=k_scale)
                    tag["applied"] = True
                else:
                    dx = conv2d_backward_data(g, w, x_shape, stride, pad, dil, groups, wparam=ctx.wsrc, k_scale=k_scale)
                    if tag is not None:
                        tag["applied"] = False
                if h is not None:
                    h["acc"] = dx
        for p_ in done:
            _sink_done(p_)
        return dx, dw, None, dshift, gres, None, None, None, None, None, None, None


class _RPNPredFn(Function):
    """The RPN head's two 1x1 predictors on every level as ONE autograd node (rpn/rpn.py:24-31):
    (cls_logits_l, bbox_pred_l) = (conv1x1(t_l, wc) + bc, conv1x1(t_l, wb) + bb).  Forward and weight / bias gradients
    are the per-level calls of _ConvFn; the data gradient of all levels and both predictors is one launch
    (cpm_rpn_pred_backward_data) that also applies the ReLU gate of the producing conv (its `gate_by_consumers` tag),
    instead of 2 x levels implicit GEMMs with reductions of 3 and 12."""

    @staticmethod
    def forward(ctx, wc, bc, wb, bb, *ts):
        H.require_gpu(wc, bc, wb, bb, *ts)
        xs = [nhwc(t) for t in ts]
        wcm, wbm = _wmem(wc), _wmem(wb)
        c = xs[0].shape[1]
        w4c, w4b = w4_of(wc, wcm, c), w4_of(wb, wbm, c)
        outs_c = [conv2d_forward(x, wcm, None, bc, None, 0, False, 1, 0, 1, 1, w4=w4c) for x in xs]
        outs_b = [conv2d_forward(x, wbm, None, bb, None, 0, False, 1, 0, 1, 1, w4=w4b) for x in xs]
        # parameters owned by the flat optimizer take their gradients in place (see _ConvFn.forward)
        ctx.sinks = [_sink_of(p, ctx.needs_input_grad[i]) if p.data_ptr() == m.data_ptr() else None
                     for i, (p, m) in enumerate(((wc, wcm), (bc, bc), (wb, wbm), (bb, bb)))]
        ctx.tags = [getattr(t, "_cpm_epi", None) for t in ts]
        ctx.n = len(ts)
        ctx.save_for_backward(wcm, wbm, *xs)
        return tuple(outs_c + outs_b)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        wcm, wbm = ctx.saved_tensors[:2]
        xs = ctx.saved_tensors[2:]
        n = ctx.n
        a, c = wcm.shape[0], wcm.shape[1]
        dcs = [nhwc(g) if g is not None else torch.zeros((x.shape[0], a, x.shape[2], x.shape[3]), device=x.device).contiguous(memory_format=CL)
               for g, x in zip(grads[:n], xs)]
        dbs = [nhwc(g) if g is not None else torch.zeros((x.shape[0], 4 * a, x.shape[2], x.shape[3]), device=x.device).contiguous(memory_format=CL)
               for g, x in zip(grads[n:], xs)]
        need = ctx.needs_input_grad
        # ---- weight / bias gradients (second stream when the parameters take them in place) ------------------------
        dw = [None, None, None, None]
        for wi, (wm, dys) in enumerate(((wcm, dcs), (wbm, dbs))):
            need_w, need_b = need[2 * wi], need[2 * wi + 1]
            if not (need_w or need_b):
                continue
            wp, bp = ctx.sinks[2 * wi], ctx.sinks[2 * wi + 1]
            if wp is not None and (bp is not None or not need_b) and need_w:
                for x, dy in zip(xs, dys):
                    if x.numel() == 0:
                        continue
                    dbias = bp._cpm_grad_sink if bp is not None else None
                    if _SIDE_WGRAD:
                        _wgrad_on_side(x, dy, wm, 1, 0, 1, 1, wp._cpm_grad_sink, dbias)
                    else:
                        conv2d_backward_weight(x, dy, wm, 1, 0, 1, 1, out=wp._cpm_grad_sink, dbias=dbias)
            else:
                # a gradient autograd accumulates: the in-place routes that were counted in forward are released
                acc_w = torch.zeros_like(wm) if need_w else None
                acc_b = torch.zeros(wm.shape[0], dtype=torch.float32, device=wm.device) if need_b else None
                for x, dy in zip(xs, dys):
                    if x.numel() == 0:
                        continue
                    if need_w:
                        conv2d_backward_weight(x, dy, wm, 1, 0, 1, 1, out=acc_w, dbias=acc_b)
                    elif need_b:
                        acc_b += dy.sum(dim=(0, 2, 3))
                dw[2 * wi], dw[2 * wi + 1] = acc_w, acc_b
                for p_ in (wp, bp):
                    if p_ is not None:
                        p_._cpm_uses -= 1
                ctx.sinks[2 * wi] = ctx.sinks[2 * wi + 1] = None
        # ---- data gradient: every level, both predictors, the producer's ReLU gate -- one launch ------------------
        dts = [None] * n
        if any(need[4:]):
            gate = all(tag is not None for tag in ctx.tags)
            dts = [torch.empty_like(x) for x in xs]
            P = H.ctypes.c_void_p
            arr = lambda ts_: (P * n)(*[t.data_ptr() for t in ts_])
            pix = (H.ctypes.c_int64 * n)(*[x.shape[0] * x.shape[2] * x.shape[3] for x in xs])
            with H.guard(xs[0].device):
                rc = H.lib().cpm_rpn_pred_backward_data(arr(dcs), arr(dbs), arr(xs), arr(dts), pix, n, H.ptr(wcm),
                                                        H.ptr(wbm), int(a), int(c), int(gate), H.stream())
            H.check(rc, "rpn_pred_backward_data")
            for tag in ctx.tags:
                if tag is not None:
                    tag["applied"] = gate
        for p_ in ctx.sinks:
            if p_ is not None:
                _sink_done(p_)
        return (dw[0], dw[1], dw[2], dw[3]) + tuple(dts)


# ---- the whole RPN head as ONE autograd node whose backward pass touches the sampled anchors only ---------------------
# rpn/rpn.py:34-41 under rpn/loss.py:88-126.  The RPN loss sums over the 256 anchors per image the sampler drew, so the
# gradient that enters the head is EXACTLY zero at every other one of its 2 x 268 569 anchors: the dense backward of the
# shared 3x3 conv (data + weight gradient over every pixel of P2..P6: ~1.4 ms of MFMA kernels per step) multiplies
# zeros.  csrc/rpn_sparse.hip has the algebra: the sampled anchors become the <= images x 256 rows of small dense
# matrices, the products run on this package's own weight- / data-gradient kernels, and the feature gradients are
# scattered into the shared accumulators.  The loss announces its sample (set_rpn_sample: an index list built on the
# device, no host round trip); without one -- another loss, a test feeding its own gradients, deterministic mode (the
# scatter uses float atomics) -- backward() runs the dense formulation.
_RPN_SPARSE = int(os.environ.get("CPM_RPN_SPARSE", "1"))
_rpn_sample = None


class _RpnToken(object):
    """one per forward pass of the RPN head"""
    __slots__ = ()


def set_rpn_sample(token, idx, cap, n_img):
    """the loss over the head outputs of the forward pass identified by `token` (the `_cpm_rpn_sparse` attribute rpn_head
    puts on every level's objectness map: one object per forward -- NOT an address, the caching allocator hands those
    out again) is a sum over the anchors listed in idx (int32 [cap], ascending flat positions, -1 behind the last):
    what _RPNHeadFn.backward may rely on"""
    global _rpn_sample
    _rpn_sample = (token, idx, int(cap), int(n_img))


def _take_rpn_sample(token):
    global _rpn_sample
    s, _rpn_sample = _rpn_sample, None
    return s if (s is not None and token is not None and s[0] is token) else None


def mask_compact(pos, neg, cap):
    """ascending positions of pos | neg (bool [T]) as int32 [cap] (-1 behind the last) + their number (int32 [1]), on the
    device (cpm_mask_compact)"""
    idx = torch.empty((cap,), dtype=torch.int32, device=pos.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=pos.device)
    ws = torch.empty((pos.numel() // 4096 + 1,), dtype=torch.int32, device=pos.device)
    with H.guard(pos.device):
        rc = H.lib().cpm_mask_compact(H.ptr(pos), H.ptr(neg), H.c_int64(pos.numel()), int(cap), H.ptr(idx), H.ptr(cnt),
                                      H.ptr(ws), H.stream())
    H.check(rc, "mask_compact")
    return idx, cnt


def _krsc_matrix(t):
    """a [K, C, R, S] weight (or gradient buffer) in KRSC memory as the [K, R*S*C, 1, 1] matrix it is (a view)"""
    k, c, r, s = t.shape
    v = t.permute(0, 2, 3, 1)
    assert v.is_contiguous()
    return v.reshape(k, r * s * c, 1, 1)


def _image_dense(g):
    """a [N, ch, H, W] gradient whose every image is a dense NHWC block (the images may lie further apart)"""
    n, ch, h, w = g.shape
    return (g.dtype == torch.float32 and g.stride(1) == 1 and g.stride(3) == ch and g.stride(2) == w * ch
            and (n == 1 or g.stride(0) >= h * w * ch))           # (one image: its stride says nothing)


def _image_stride(g):
    """floats between two images of an _image_dense tensor (a single image: the dense distance)"""
    return int(g.stride(0)) if g.shape[0] > 1 else int(g.shape[1] * g.shape[2] * g.shape[3])


class _RPNHeadFn(Function):
    """(objectness_l, deltas_l for every level l) = RPNHead(features): shared 3x3 conv + ReLU, the two 1x1 predictors."""

    @staticmethod
    def forward(ctx, w, b, wc, bc, wb, bb, *feats):
        H.require_gpu(w, b, wc, bc, wb, bb, *feats)
        xs = [nhwc(f) for f in feats]
        wm, wcm, wbm = _wmem(w), _wmem(wc), _wmem(wb)
        c = xs[0].shape[1]
        w4, w4c, w4b = w4_of(w, wm, c), w4_of(wc, wcm, c), w4_of(wb, wbm, c)
        # the shared conv on the coarser levels runs on the second stream beside the finest level's (fwd_fork)
        forked = len(xs) > 1 and fwd_fork(xs[0], 4)
        ts = []
        for li, x in enumerate(xs):
            if forked and li > 0:
                with fwd_side(x):
                    ts.append(conv2d_forward(x, wm, None, b, None, 0, True, 1, 1, 1, 1, w4=w4))
            else:
                ts.append(conv2d_forward(x, wm, None, b, None, 0, True, 1, 1, 1, 1, w4=w4))
        if forked:
            fwd_join(xs[0])
        outs_c = [conv2d_forward(t, wcm, None, bc, None, 0, False, 1, 0, 1, 1, w4=w4c) for t in ts]
        outs_b = [conv2d_forward(t, wbm, None, bb, None, 0, False, 1, 0, 1, 1, w4=w4b) for t in ts]
        # parameters owned by the flat optimizer take their gradients in place (see _ConvFn.forward)
        ctx.sinks = [_sink_of(p, ctx.needs_input_grad[i]) if (p.data_ptr() == m.data_ptr() and getattr(
            p, "_cpm_grad_sink", None) is not None and p._cpm_grad_sink.stride() == p.stride()) else None
            for i, (p, m) in enumerate(((w, wm), (b, b), (wc, wcm), (bc, bc), (wb, wbm), (bb, bb)))]
        ctx.holders = [getattr(f, "_cpm_gacc", None) for f in feats]
        ctx.n = len(xs)
        ctx.token = _RpnToken()                     # identifies THIS forward pass to the loss (set_rpn_sample)
        ctx.save_for_backward(wm, wcm, wbm, *xs, *ts)
        return tuple(outs_c + outs_b)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        if H.deferred:
            H.run_deferred()
        n = ctx.n
        wm, wcm, wbm = ctx.saved_tensors[:3]
        xs, ts = ctx.saved_tensors[3:3 + n], ctx.saved_tensors[3 + n:3 + 2 * n]
        a, c, k = wcm.shape[0], wcm.shape[1], wm.shape[0]
        dev = xs[0].device
        need = ctx.needs_input_grad

        def zmap(x, ch):
            return torch.zeros((x.shape[0], ch, x.shape[2], x.shape[3]), device=dev).contiguous(memory_format=CL)
        # (the sparse path reads <= cap elements of these maps: it takes them as they come -- slices of the loss's one
        # gradient tensor, dense inside an image -- instead of paying ten transposing copies; see _image_dense)
        sample = _take_rpn_sample(ctx.token)
        sparse = sample is not None and _RPN_SPARSE and (_RPN_SPARSE == 2 or not H.deterministic())
        raw = sparse and all(g is not None and _image_dense(g) for g in grads[:2 * n])
        dcs = [g if raw else (nhwc(g) if g is not None else zmap(x, a)) for g, x in zip(grads[:n], xs)]
        dbs = [g if raw else (nhwc(g) if g is not None else zmap(x, 4 * a)) for g, x in zip(grads[n:], xs)]
        # gradient buffers of the six parameters: the flat optimizer's slice (accumulated in place) or a fresh one
        bufs, rets = [], []
        for i, like in enumerate((wm, None, wcm, None, wbm, None)):
            p = ctx.sinks[i]
            if not need[i]:
                bufs.append(None); rets.append(None)
            elif p is not None:
                bufs.append(p._cpm_grad_sink); rets.append(None)
            else:
                z = torch.zeros_like(like) if like is not None else torch.zeros((k, a, 4 * a)[i // 2], device=dev)
                bufs.append(z); rets.append(z)
        # gradient tensors of the feature maps: the shared accumulator when another consumer already made one
        dfe, dret = [], []
        for i, (x, h) in enumerate(zip(xs, ctx.holders)):
            if not need[6 + i]:
                dfe.append(None); dret.append(None)
            elif h is not None and "acc" in h and tuple(h["acc"].shape) == tuple(x.shape):
                _wait_readers(h)
                dfe.append(h["acc"]); dret.append(None)
            else:
                dfe.append(False); dret.append(None)        # made below (zero-filled / fresh), the two paths differ
        # (_RPN_SPARSE == 2: also under deterministic mode -- the tests compare the two formulations there, where
        # everything else of a step is reproducible)
        if sparse:
            _RPNHeadFn._backward_sparse(ctx, sample, dcs, dbs, xs, ts, wm, wcm, wbm, bufs, dfe, dret)
        else:
            _RPNHeadFn._backward_dense(ctx, dcs, dbs, xs, ts, wm, wcm, wbm, bufs, dfe, dret)
        for p_ in ctx.sinks:
            if p_ is not None:
                _sink_done(p_)
        return tuple(rets) + tuple(dret)

    @staticmethod
    def _own(ctx, i, x, dfe, dret, zero):
        """a gradient tensor of feature map i that this node makes itself (nobody accumulated into one yet)"""
        t = torch.zeros_like(x) if zero else torch.empty_like(x)
        h = ctx.holders[i]
        if h is not None:
            h["acc"] = t
        dfe[i], dret[i] = t, t
        return t

    @staticmethod
    def _backward_sparse(ctx, sample, dcs, dbs, xs, ts, wm, wcm, wbm, bufs, dfe, dret):
        _, idx, cap, n_img = sample
        n, dev = ctx.n, xs[0].device
        a, c, k = wcm.shape[0], wcm.shape[1], wm.shape[0]
        DT = torch.empty((cap, k), dtype=torch.float32, device=dev)
        Gc = torch.empty((cap, a), dtype=torch.float32, device=dev)
        Gb = torch.empty((cap, 4 * a), dtype=torch.float32, device=dev)
        T = torch.empty((cap, c), dtype=torch.float32, device=dev)
        X = torch.empty((cap, 9 * c), dtype=torch.float32, device=dev)
        pix = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        P_ = H.ctypes.c_void_p
        arr = lambda ts_: (P_ * n)(*[t.data_ptr() if t is not None else None for t in ts_])
        hs = (H.ctypes.c_int * n)(*[int(x.shape[2]) for x in xs])
        ws = (H.ctypes.c_int * n)(*[int(x.shape[3]) for x in xs])
        i64 = H.ctypes.c_int64 * n
        with H.guard(dev):
            rc = H.lib().cpm_rpn_sparse_rows(H.ptr(idx), cap, n_img, n, hs, ws, int(a), int(c), arr(dcs), arr(dbs),
                                             arr(ts), arr(xs), H.ptr(wcm), H.ptr(wbm), H.ptr(DT), H.ptr(Gc), H.ptr(Gb),
                                             H.ptr(T), H.ptr(X), H.ptr(pix), i64(*[_image_stride(g) for g in dcs]),
                                             i64(*[_image_stride(g) for g in dbs]), H.stream())
        H.check(rc, "rpn_sparse_rows")
        img = lambda m: m.view(1, cap, 1, m.shape[1]).permute(0, 3, 1, 2)       # [1, channels, rows, 1], NHWC memory
        # the 3x3 conv: dW = DT^T X (+ the bias sum), the predictors: dWcls = Gc^T T, dWbox = Gb^T T
        for wi, (xm, gm, wlike) in enumerate(((X, DT, wm), (T, Gc, wcm), (T, Gb, wbm))):
            dwb, dbb = bufs[2 * wi], bufs[2 * wi + 1]
            if dwb is not None:
                out = _krsc_matrix(dwb)
                conv2d_backward_weight(img(xm), img(gm), out, 1, 0, 1, 1, out=out, dbias=dbb)
            elif dbb is not None:
                dbb += gm.sum(0)
        # the feature maps: dX = DT W, scattered to the nine input pixels of every row
        if any(d is not None for d in dfe):
            for i, x in enumerate(xs):
                if dfe[i] is False:
                    _RPNHeadFn._own(ctx, i, x, dfe, dret, zero=True)
            dX = conv2d_backward_data(img(DT), _krsc_matrix(wm), (1, 9 * c, cap, 1), 1, 0, 1, 1)
            with H.guard(dev):
                rc = H.lib().cpm_rpn_sparse_scatter(H.ptr(pix), cap, n, hs, ws, int(c), H.ptr(dX), arr(dfe), H.stream())
            H.check(rc, "rpn_sparse_scatter")

    @staticmethod
    def _backward_dense(ctx, dcs, dbs, xs, ts, wm, wcm, wbm, bufs, dfe, dret):
        """every pixel of every level (what autograd would do through the three convolutions)"""
        n, dev = ctx.n, xs[0].device
        a, c = wcm.shape[0], wcm.shape[1]
        # predictors: weight / bias gradients per level, then their data gradient with the conv's ReLU gate
        for wi, (wlike, dys) in enumerate(((wcm, dcs), (wbm, dbs))):
            dwb, dbb = bufs[2 + 2 * wi], bufs[3 + 2 * wi]
            for t, dy in zip(ts, dys):
                if t.numel() == 0:
                    continue
                if dwb is not None:
                    conv2d_backward_weight(t, dy, wlike, 1, 0, 1, 1, out=dwb, dbias=dbb)
                elif dbb is not None:
                    dbb += dy.sum(dim=(0, 2, 3))
        dts = [torch.empty_like(t) for t in ts]
        P_ = H.ctypes.c_void_p
        arr = lambda ts_: (P_ * n)(*[t.data_ptr() for t in ts_])
        pix = (H.ctypes.c_int64 * n)(*[t.shape[0] * t.shape[2] * t.shape[3] for t in ts])
        with H.guard(dev):
            rc = H.lib().cpm_rpn_pred_backward_data(arr(dcs), arr(dbs), arr(ts), arr(dts), pix, n, H.ptr(wcm),
                                                    H.ptr(wbm), int(a), int(c), 1, H.stream())
        H.check(rc, "rpn_pred_backward_data")
        for i, (x, dt) in enumerate(zip(xs, dts)):
            if x.numel() == 0:
                continue
            if bufs[0] is not None:
                conv2d_backward_weight(x, dt, wm, 1, 1, 1, 1, out=bufs[0], dbias=bufs[1])
            elif bufs[1] is not None:
                bufs[1] += dt.sum(dim=(0, 2, 3))
            if dfe[i] is None:
                continue
            if dfe[i] is False:
                t = conv2d_backward_data(dt, wm, tuple(x.shape), 1, 1, 1, 1)
                h = ctx.holders[i]
                if h is not None:
                    h["acc"] = t
                dfe[i], dret[i] = t, t
            else:
                conv2d_backward_data(dt, wm, tuple(x.shape), 1, 1, 1, 1, accumulate_into=dfe[i])


def rpn_head(feats, w, b, wc, bc, wb, bb):
    """([objectness(f) for f in feats], [deltas(f) for f in feats]) of the RPN head as one autograd node (_RPNHeadFn) whose
    backward pass is sparse under the RPN loss; None when the shapes are outside its kernels (the caller then goes conv
    by conv)."""
    c = wc.shape[1]
    if (not _RPN_SPARSE or len(feats) < 1 or len(feats) > 8 or c % 4 or 256 % (c // 4) or 5 * wc.shape[0] * c * 4 > 65536
            or wb.shape[0] != 4 * wc.shape[0] or tuple(wc.shape[2:]) != (1, 1) or tuple(wb.shape[2:]) != (1, 1)
            or tuple(w.shape) != (c, c, 3, 3) or b is None or bc is None or bb is None
            or any(t.dim() != 4 or t.shape[1] != c for t in feats) or not torch.is_grad_enabled()):
        return None
    global _rpn_sample
    _rpn_sample = None                  # a sample announced for an earlier forward whose backward never ran is void
    del H.deferred[:]                   # (likewise work a failed backward pass left parked)
    outs = _RPNHeadFn.apply(w, b, wc, bc, wb, bb, *feats)
    n = len(feats)
    lo, br = list(outs[:n]), list(outs[n:])
    tok = getattr(outs[0].grad_fn, "token", None)      # (a Function's ctx IS its outputs' grad_fn)
    for o in lo:
        o._cpm_rpn_sparse = tok         # the loss that sums over a sample of these anchors may say so (set_rpn_sample)
    return lo, br


_RPN_PRED_FUSED = os.environ.get("CPM_RPN_PRED_FUSED", "1") != "0"


def rpn_predictors(ts, wc, bc, wb, bb):
    """([cls_logits(t) for t in ts], [bbox_pred(t) for t in ts]) -- one autograd node, see _RPNPredFn.  None when the
    shapes are outside the fused data-gradient kernel (the caller then runs the two convs per level)."""
    c = wc.shape[1]
    if (not _RPN_PRED_FUSED or len(ts) < 1 or len(ts) > 8 or c % 4 or 256 % (c // 4) or 5 * wc.shape[0] * c * 4 > 65536
            or wb.shape[0] != 4 * wc.shape[0] or tuple(wc.shape[2:]) != (1, 1) or tuple(wb.shape[2:]) != (1, 1)
            or bc is None or bb is None or any(t.dim() != 4 or t.shape[1] != c for t in ts)):
        return None
    outs = _RPNPredFn.apply(wc, bc, wb, bb, *ts)
    n = len(ts)
    return list(outs[:n]), list(outs[n:])


def conv2d(x, w, scale=None, shift=None, residual=None, stride=1, pad=0, dil=1, groups=1, relu=False, res_mode=0,
           sole_consumer=False, gate_by_consumers=False):
    """sole_consumer / gate_by_consumers: the caller promises that y = relu(..) is consumed ONLY by convolutions of
    this package (conv2d / linear; as their input, or as the residual of one) -- they then apply y's ReLU gate inside
    their data-gradient kernels (see _ConvFn.forward) and this layer's backward needs no elementwise pass.
    sole_consumer: exactly one consumer (bottleneck conv1 -> conv2 -> conv3); gate_by_consumers: several (a bottleneck's
    output: the next block's conv1, its downsample conv or residual add, an FPN lateral)."""
    tag = None
    if not torch.is_grad_enabled():
        # inference: no autograd node, no context (the test-time forward is bound by the host: ~100 conv calls per image)
        H.require_gpu(x, w, scale, shift, residual)
        xn, wm = nhwc(x), _wmem(w)
        res = nhwc(residual) if residual is not None else None
        _side_copies((x, xn), (w, wm), (residual, res))
        return conv2d_forward(xn, wm, scale, shift, res, res_mode, relu, stride, pad, dil, groups,
                              w4=w4_of(w, wm, xn.shape[1] // groups))
    if gate_by_consumers and not _GATE_BY_CONSUMERS:
        gate_by_consumers = False
    if relu and ((sole_consumer and residual is None) or gate_by_consumers):
        tag = {"applied": False}
    y = _ConvFn.apply(x, w, scale, shift, residual, stride, pad, dil, groups, relu, res_mode, tag)
    if tag is not None:
        y._cpm_epi = tag
    return y


def carry_tag(src, dst):
    """reshape / view make a new tensor object: hand the sole-consumer tag (conv2d) on to it."""
    tag = getattr(src, "_cpm_epi", None)
    if tag is not None:
        dst._cpm_epi = tag
    return dst


def linear(x, w, bias=None, relu=False, sole_consumer=False):
    """nn.Linear as a 1x1 conv on a 1x1 image: x [R, C], w [K, C].  sole_consumer: see conv2d (the caller promises that
    the result feeds exactly one further conv2d / linear of this package)."""
    r, c = x.shape
    w4 = w.reshape(w.shape[0], c, 1, 1)
    sink = getattr(w, "_cpm_grad_sink", None)
    if sink is not None and w4.data_ptr() == w.data_ptr() and sink.is_contiguous():
        # the view stands in for the parameter: the weight-gradient kernel accumulates into the parameter's slice of
        # the flat gradient buffer and the data-parallel reducer hears about it (see _ConvFn.forward)
        # uses are counted on the PARAMETER (a fresh view per call would fire the reducer's ready hook once per use
        # instead of once per step when a Linear is applied more than once)
        w4._cpm_grad_sink = sink.view(w4.shape)
        w4._cpm_owner = w
    y = conv2d(carry_tag(x, x.reshape(r, c, 1, 1)), w4, None, bias, None, relu=relu, sole_consumer=sole_consumer)
    return carry_tag(y, y.reshape(r, w.shape[0]))


def upsample2x_add_backward(dy, top_shape):
    n, c, p, q = dy.shape
    assert tuple(top_shape) == (n, c, (p + 1) // 2, (q + 1) // 2), "top-down sizes must be 2x apart"
    dtop = empty_nhwc(top_shape, dy)
    with H.guard(dy.device):
        rc = H.lib().cpm_upsample2x_add_backward(H.ptr(nhwc(dy)), n, p, q, c, H.ptr(dtop), 0, H.stream())
    H.check(rc, "upsample2x_add_backward")
    return dtop


class _ConvTransposeFn(Function):
    """nn.ConvTranspose2d (+bias, +ReLU).  w: torch layout [Cin, Cout/groups, R, S], KRSC memory."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, groups, relu):
        H.require_gpu(x, w, bias)
        x = nhwc(x)
        w_in = w
        w = _wmem(w)
        # parameters owned by the flat optimizer: gradients are accumulated in place (see _ConvFn.forward)
        sink = getattr(w_in, "_cpm_grad_sink", None)
        ctx.wparam = w_in if (ctx.needs_input_grad[1] and w is w_in and sink is not None and sink.data_ptr() != 0
                              and sink.stride() == w.stride()) else None
        if ctx.wparam is not None:
            _note_use(w_in)
        ctx.bparam = _sink_of(bias, bias is not None and ctx.needs_input_grad[2])
        n, cin, p, q = x.shape
        _, cog, r, s = w.shape
        cout = cog * groups
        hh, ww = (p - 1) * stride - 2 * pad + r, (q - 1) * stride - 2 * pad + s
        d = make_desc(n, cout, hh, ww, cin, r, s, stride, pad, 1, groups)
        assert d.P == p and d.Q == q
        y = empty_nhwc((n, cout, hh, ww), x)
        if y.numel():
            ws = _ws(d, x.device)
            with H.guard(x.device):
                rc = H.lib().cpm_conv_transpose2d_forward(H.ctypes.byref(d), H.ptr(x), H.ptr(w), H.ptr(bias),
                                                          int(bool(relu)), H.ptr(y), H.ptr(ws),
                                                          H.c_size_t(ws.numel()), H.stream())
            H.check(rc, "conv_transpose2d_forward")
        ctx.cfg = (stride, pad, groups, relu, bias is not None)
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, groups, relu, has_bias = ctx.cfg
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dy = nhwc(dy)
        dbias = None
        if relu or (has_bias and need_b):
            bp = ctx.bparam
            dpre, _, dbias = epilogue_backward(dy, y, None, relu, want_dpre=relu, want_dshift=has_bias and need_b,
                                               dshift_out=bp._cpm_grad_sink if bp is not None else None)
            if bp is not None:
                dbias = None                    # accumulated in place
                _sink_done(bp)
            dy = dpre if dpre is not None else dy
        # the transposed conv's data gradient is the plain conv; its weight gradient swaps the operands
        dx = conv2d_forward(dy, w, None, None, None, 0, False, stride, pad, 1, groups) if need_x else None
        dw = None
        if need_w:
            wp = ctx.wparam
            if wp is not None:
                conv2d_backward_weight(dy, x, w, stride, pad, 1, groups, out=wp._cpm_grad_sink)
                _sink_done(wp)
            else:
                dw = conv2d_backward_weight(dy, x, w, stride, pad, 1, groups)
        return dx, dw, dbias, None, None, None, None


def conv_transpose2d(x, w, bias=None, stride=2, pad=1, groups=1, relu=False):
    return _ConvTransposeFn.apply(x, w, bias, stride, pad, groups, relu)


class _GroupNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        H.require_gpu(x, gamma, beta)
        x = nhwc(x)
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((n, groups), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if n:
            with H.guard(x.device):
                rc = H.lib().cpm_groupnorm_forward(H.ptr(x), H.ptr(gamma), H.ptr(beta), n, h * w, c, int(groups),
                                                   H.f(eps), int(bool(relu)), H.ptr(y), H.ptr(mean), H.ptr(rstd),
                                                   H.stream())
            H.check(rc, "groupnorm_forward")
        ctx.cfg = (groups, relu)
        ctx.gparam = _sink_of(gamma, ctx.needs_input_grad[1])
        ctx.bparam = _sink_of(beta, ctx.needs_input_grad[2])
        ctx.save_for_backward(x, y, gamma, mean, rstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, gamma, mean, rstd = ctx.saved_tensors
        groups, relu = ctx.cfg
        dy = nhwc(dy)
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        gp, bp = ctx.gparam, ctx.bparam
        dgamma = gp._cpm_grad_sink if gp is not None else torch.zeros_like(gamma)     # the kernel accumulates
        dbeta = bp._cpm_grad_sink if bp is not None else torch.zeros_like(gamma)
        if n:
            with H.guard(x.device):
                rc = H.lib().cpm_groupnorm_backward(H.ptr(dy), H.ptr(x), H.ptr(y), H.ptr(gamma), H.ptr(mean),
                                                    H.ptr(rstd), n, h * w, c, int(groups), int(bool(relu)),
                                                    H.ptr(dx), H.ptr(dgamma), H.ptr(dbeta), H.stream())
            H.check(rc, "groupnorm_backward")
        if gp is not None:
            dgamma = None
            _sink_done(gp)
        if bp is not None:
            dbeta = None
            _sink_done(bp)
        return dx, dgamma, dbeta, None, None, None


def group_norm(x, gamma, beta, groups, eps=1e-5, relu=False):
    return _GroupNormFn.apply(x, gamma, beta, groups, eps, relu)


# ---- a stack of [conv + bias -> GroupNorm -> ReLU] layers as ONE autograd node (cpm_layer_chain_*) -------------
_STACK = os.environ.get("CPM_CONV_GN_STACK", "1") != "0"


class _ChainPlan(object):
    """What does not change from call to call about a chain of [conv + bias -> (GroupNorm) -> (ReLU)] layers on a
    given per-sample input shape: the native layer table (cpm_chain_layer[]: per-sample geometry, parameter and
    gradient-sink pointers).  Built once; the number of samples (RoIs) varies from step to step and only sizes the two
    buffers of a call (cpm_layer_chain_sizes, remembered per N).  The chains this exists for run ~0.3 ms of kernels:
    dozens of small allocations or ~150 ctypes field stores in front of the first launch would cost what the native
    loop saves."""

    def __init__(self, chw, specs, infer=False):
        """specs: per layer (weight [K,C,R,S] or [K,C], bias, gamma or None, beta or None, stride, pad, gn_groups, eps,
        relu).  infer: a forward-only table (test time: parameters without gradient sinks)"""
        self.infer = infer
        sink = (lambda t: 0) if infer else (lambda t: t._cpm_grad_sink.data_ptr())
        self.n = len(specs)
        self.table = (H.ChainLayer * self.n)()
        params, self.dgrad_keys = [], []
        c, h, wd = chw
        for i, (w, bias, gamma, beta, stride, pad, gn_groups, eps, relu) in enumerate(specs):
            k = w.shape[0]
            r, s = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
            L = self.table[i]
            L.conv = make_desc(1, c, h, wd, k, r, s, stride, pad, 1, 1)
            L.w, L.bias = w.data_ptr(), bias.data_ptr()
            L.dw, L.dbias = sink(w), sink(bias)
            L.has_gn, L.relu = int(gamma is not None), int(bool(relu))
            layer_params = [w, bias]
            if gamma is not None:
                L.gamma, L.beta = gamma.data_ptr(), beta.data_ptr()
                L.dgamma, L.dbeta = sink(gamma), sink(beta)
                L.gn_groups, L.eps = int(gn_groups), float(eps)
                layer_params += [gamma, beta]
            params += layer_params
            # conv2d_backward_data's rule: a full-window layer (fc6, iou_fc1) runs its data gradient as the 1x1 problem
            # over R*S*C channels it is, with the weight registered as that [K, R*S*C] matrix
            full_window = (r, s) == (h, wd) and pad == 0 and stride == 1 and (r > 1 or s > 1)
            L.dgrad_flat = int(full_window)
            self.dgrad_keys.append((w, 1, k, 1, r * s * c) if full_window else (w, 1, k, r * s, c))
            c, h, wd = k, L.conv.P, L.conv.Q
        self.out_chw = (c, h, wd)
        self.flat_out = specs[-1][0].dim() == 2                  # a Linear ends the chain: [N, K] output
        self.params = tuple(params)
        self.ptrs = self._pointers()
        self.wt_ptrs = None
        self.w4_ptrs = None
        self.w4_key = None
        self.weights = tuple(sp[0] for sp in specs)
        self.sizes = {}

    def _pointers(self):
        if self.infer:
            return tuple(p.data_ptr() for p in self.params)
        return tuple(p.data_ptr() for p in self.params) + tuple(p._cpm_grad_sink.data_ptr() for p in self.params)

    def stale(self):
        """a parameter or its gradient sink moved (load_state_dict into new storage, a rebuilt optimizer)"""
        return self.ptrs != self._pointers()

    def sizes_for(self, n):
        r = self.sizes.get(n)
        if r is None:
            f, b, w = H.c_size_t(), H.c_size_t(), H.c_size_t()
            H.check(H.lib().cpm_layer_chain_sizes(self.table, self.n, int(n), H.ctypes.byref(f), H.ctypes.byref(b),
                                                 H.ctypes.byref(w)), "layer_chain_sizes")
            if len(self.sizes) > 1024:
                self.sizes.clear()
            r = self.sizes[n] = (int(f.value), int(b.value), int(w.value))
        return r

    def refresh_weight_images(self):
        """data-gradient weight images (FlatSGD._refresh_dgrad_weights): registered on first use, present from the
        first optimizer step on; their addresses then stay put"""
        wts = []
        for key in self.dgrad_keys:
            wt = _prepared_wt(*key)
            wts.append(wt.data_ptr() if wt is not None else None)
            wts.append(int(getattr(key[0], "_cpm_wt_fmt", 0)) if wt is not None else 0)
        wts = tuple(wts)
        if wts != self.wt_ptrs:
            for i in range(self.n):
                self.table[i].wt, self.table[i].wt_w4 = wts[2 * i], wts[2 * i + 1]
            self.wt_ptrs = wts

    def _w4_state(self):
        """(version, version of the image, address of the image) per weight: equal states mean the table is current"""
        out = []
        for w in self.weights:
            own = getattr(w, "_cpm_owner", w)
            t = getattr(own, "_cpm_w4", None)
            out.append((own._version, getattr(own, "_cpm_w4_version", None), t.data_ptr() if t is not None else 0))
        return tuple(out)

    def refresh_w4(self):
        """pre-split images of the weights for the forward convs (w4_of); the native loop reads them under bf16x3 only"""
        if not bf16x3():
            return
        if self.w4_key is not None and self.w4_key == self._w4_state():
            return                                      # nothing moved since the table was filled
        ptrs = []
        for w in self.weights:
            t = w4_of(w, w if w.dim() == 2 else _wmem(w), w.shape[1])
            ptrs.append(t.data_ptr() if t is not None else None)
        ptrs = tuple(ptrs)
        if ptrs != self.w4_ptrs:
            for i, v in enumerate(ptrs):
                self.table[i].w4 = v
            self.w4_ptrs = ptrs
        state = self._w4_state()
        self.w4_key = state if all(v == iv for v, iv, _ in state) else None


def _chain_forward(x, plan):
    """the native forward loop of a chain (cpm_layer_chain_forward): (input as read, activation buffer, output)"""
    H.require_gpu(x)
    H.wait_pending_sgd(x.device)            # (the chain's parameters travel as raw pointers in its plan)
    x_in = x
    x = nhwc(x) if x.dim() == 4 else x.contiguous()
    _side_copies((x_in, x))
    n = x.shape[0]
    fwd_floats, _, ws_bytes = plan.sizes_for(n)
    fbuf = H.side_alloc(lambda: torch.empty((fwd_floats,), dtype=torch.float32, device=x.device))
    if plan.flat_out:
        y = H.side_alloc(lambda: torch.empty((n, plan.out_chw[0]), dtype=torch.float32, device=x.device))
    else:
        y = H.side_alloc(lambda: torch.empty((n,) + plan.out_chw, dtype=torch.float32, device=x.device,
                                             memory_format=CL))
    if not plan.infer:
        for p_ in plan.params:
            _note_use(p_)
    plan.refresh_w4()
    ws = H.workspace(ws_bytes, x.device)
    with H.guard(x.device):
        rc = H.lib().cpm_layer_chain_forward(plan.table, plan.n, n, H.ptr(x), H.ptr(fbuf), H.ptr(y), H.ptr(ws),
                                             H.c_size_t(ws.numel()), H.stream())
    H.check(rc, "layer_chain_forward")
    return x, fbuf, y


class _LayerChainFn(Function):
    """x -> L x [conv(w, b) -> (GroupNorm) -> (ReLU)]: the calls of _ConvFn / _GroupNormFn layer by layer, issued by one
    C loop per direction (cpm_layer_chain_*).  The parameters are not inputs of the node: each owns a slice of the flat
    gradient buffer that the kernels accumulate into, and the reducer is told directly."""

    @staticmethod
    def forward(ctx, x, plan):
        x, fbuf, y = _chain_forward(x, plan)
        ctx.plan = plan
        ctx.save_for_backward(x, fbuf, y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, fbuf, y = ctx.saved_tensors
        plan = ctx.plan
        dy = nhwc(dy) if dy.dim() == 4 else dy.contiguous()
        dev = x.device
        n = x.shape[0]
        _, bwd_floats, ws_bytes = plan.sizes_for(n)
        bbuf = torch.empty((bwd_floats,), dtype=torch.float32, device=dev)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        plan.refresh_weight_images()
        ws = H.workspace(ws_bytes, dev)
        side_raw, side_ws, st = None, None, None
        if _SIDE_WGRAD:
            wgrad_stream(dev)
            st = _side[dev.index]
            side_raw = st[1]
            with H.use_stream(side_raw):
                side_ws = H.workspace(ws_bytes, dev)
        main_raw = H._raw_stream(dev.index)
        with H.guard(dev):
            rc = H.lib().cpm_layer_chain_backward(plan.table, plan.n, n, H.ptr(x), H.ptr(dy), H.ptr(fbuf), H.ptr(y),
                                                  H.ptr(bbuf), H.ptr(dx), H.ptr(ws), H.c_size_t(ws.numel()),
                                                  H.ptr(side_ws),
                                                  H.c_size_t(side_ws.numel() if side_ws is not None else 0),
                                                  H.c_void_p(main_raw), H.c_void_p(side_raw))
        H.check(rc, "layer_chain_backward")
        if side_raw is not None:
            # the weight gradients read the layer inputs (x, the outputs inside fbuf), dy and d_conv (bbuf) over there
            for t in (x, fbuf, bbuf, dy):
                t.record_stream(st[0])
            _side_armed[dev.index] = main_raw
            torch.autograd.Variable._execution_engine.queue_callback(_join_side)
        for p_ in plan.params:
            _sink_done(p_)
        return dx, None


def _sunk(p):
    return p is not None and p.requires_grad and getattr(p, "_cpm_grad_sink", None) is not None \
        and p._cpm_grad_sink.data_ptr() != 0


def layer_chain(x, layers, owner):
    """Run `layers` = [(conv_or_linear_module, GroupNorm_module_or_None, relu), ...] on x as ONE autograd node with one
    native call per direction (cpm_layer_chain_*) when training on the flat gradient buffer -- and, forward only, under
    no_grad (the test-time forward is bound by the host: 16 op calls per grid stage become one); returns None when the
    chain does not qualify (the caller then goes layer by layer).  `owner`: the module the plans are cached on."""
    infer = not torch.is_grad_enabled()     # test time: the same native loop, forward only, no gradient sinks needed
    if not (_STACK and x.is_cuda and (infer or x.requires_grad) and x.shape[0] > 0):
        return None
    if infer and os.environ.get("CPM_CHAIN_INFER", "1") == "0":
        return None
    cache = owner.__dict__.setdefault("_cpm_chain_plans", {})
    chw = tuple(x.shape[1:]) if x.dim() == 4 else (x.shape[1], 1, 1)
    key = (chw, len(layers), infer)
    plan = cache.get(key)
    if plan is not None and plan.stale():
        plan = None
    if plan is None:
        specs = []
        rc, rh, rw = chw                                 # the shape running through the chain: the native table only
        for i, (m, gn, relu) in enumerate(layers):       # holds pointers, so every layer is checked against it HERE
            w, b = m.weight, m.bias
            if b is None or (gn is not None and gn.bias is None):
                return None
            if not infer and (not (_sunk(w) and _sunk(b)) or (gn is not None and not (_sunk(gn.weight) and _sunk(gn.bias)))):
                return None
            if w.dim() == 4:
                conv_like = hasattr(m, "stride")
                stride, pad = (m.stride[0], m.padding[0]) if conv_like else (1, 0)
                if conv_like and (m.groups != 1 or m.dilation[0] != 1 or m.in_channels <= 1 or m.padding_mode != "zeros"):
                    return None
                if not w.is_contiguous(memory_format=CL):
                    return None
                if w.shape[1] != rc:
                    return None
                if not conv_like:
                    # a windowed Linear (a full-window conv): its window must BE the running map
                    if getattr(m, "window", None) != (rc, rh, rw) or (w.shape[2], w.shape[3]) != (rh, rw):
                        return None
                rh, rw = out_size(rh, w.shape[2], stride, pad), out_size(rw, w.shape[3], stride, pad)
                if rh < 1 or rw < 1:
                    return None
                rc = w.shape[0]
            else:
                stride, pad = 1, 0
                if not (w.dim() == 2 and w.is_contiguous()):
                    return None
                if (rh, rw) != (1, 1) or w.shape[1] != rc:   # a flat Linear behind a map with H, W > 1 is NOT a 1x1 conv
                    return None
                rc = w.shape[0]
            if gn is not None and gn.num_channels != rc:
                return None
            if gn is None and relu and i == len(layers) - 1:
                return None                                  # a bare ReLU at the end needs its consumer's gate
            specs.append((w, b, gn.weight if gn is not None else None, gn.bias if gn is not None else None, stride, pad,
                          gn.num_groups if gn is not None else 0, gn.eps if gn is not None else 0.0, relu))
        if len(cache) > 16:
            cache.clear()
        plan = cache[key] = _ChainPlan(chw, specs, infer)
    if infer:
        return _chain_forward(x, plan)[2]
    return _LayerChainFn.apply(x, plan)


def conv_gn_stack(x, convs, norms):
    """[conv(x) -> GroupNorm -> ReLU] over lists of ops.Conv2d / ops.GroupNorm modules (the grid head): one autograd
    node when it qualifies (layer_chain), layer by layer otherwise."""
    y = layer_chain(x, [(cv, gn, True) for cv, gn in zip(convs, norms)], convs[0]) if x.dim() == 4 else None
    if y is not None:
        return y
    for cv, gn in zip(convs, norms):
        x = gn(cv(x), relu=True)
    return x


def mlp_chain(x, linears, owner):
    """Linear -> ReLU -> ... -> Linear (no ReLU after the last one) over ops.Linear modules -- fc6 / fc7 / cls_score,
    iou_fc1 / iou_fc2 / iou_pred: one autograd node when it qualifies, the per-module calls (with the consumer-side
    ReLU gates of conv2d / linear) otherwise."""
    last = len(linears) - 1
    y = layer_chain(x, [(m, None, i != last) for i, m in enumerate(linears)], owner)
    if y is not None:
        return y
    for i, m in enumerate(linears):
        x = m(x, relu=True, sole_consumer=True) if i != last else m(x)
    return x


_STEM_FUSED = os.environ.get("CPM_STEM_FUSED", "1") != "0"


def stem_forward(x, w_pad, scale, shift, r=7, s=7, stride=2, pad=3, w=None):
    """ResNet stem (backbone/ResNet.py:123-136): 7x7/s2 conv on 3 channels with the frozen affine + ReLU fused, then the
    3x3/s2 max-pool.  Frozen stage: forward only.  Under bf16x3 on an NHWC image the conv is ONE kernel reading the
    image (cpm_stem7x7_forward, `w` = conv1.weight); otherwise im2col + 1x1 MFMA GEMM on
    w_pad: [K, Kpad, 1, 1] = conv1.weight in (r,s,c) column order, zero padded to a multiple of 32."""
    H.require_gpu(x, w_pad, scale, shift)
    n, c, h, wd = x.shape
    is_nhwc = x.is_contiguous(memory_format=CL) and not x.is_contiguous()
    p, q = out_size(h, r, stride, pad), out_size(wd, s, stride, pad)
    if (_STEM_FUSED and w is not None and is_nhwc and (c, r, s, stride, pad) == (3, 7, 7, 2, 3) and w.shape[0] == 64
            and bf16x3() and n * h * wd * 3 < 2 ** 31):
        y = empty_nhwc((n, 64, p, q), x)
        with H.guard(x.device):
            rc = H.lib().cpm_stem7x7_forward(H.ptr(x), H.ptr(_wmem(w.detach())), H.ptr(scale), H.ptr(shift), 1, n, h, wd,
                                             H.ptr(y), H.stream())
        H.check(rc, "stem7x7_forward")
    else:
        xin = x if (is_nhwc or x.is_contiguous()) else x.contiguous()
        kpad = w_pad.shape[1]
        cols = torch.empty((n * p * q, kpad), dtype=torch.float32, device=x.device)
        with H.guard(x.device):
            rc = H.lib().cpm_im2col(H.ptr(xin), 1 if is_nhwc else 0, n, c, h, wd, r, s, stride, pad, p, q, kpad,
                                    H.ptr(cols), H.stream())
        H.check(rc, "im2col")
        y = conv2d_forward(cols.view(n, p, q, kpad).permute(0, 3, 1, 2), w_pad, scale, shift, None, 0, True, 1, 0, 1, 1)
    pp, pq = out_size(p, 3, 2, 1), out_size(q, 3, 2, 1)
    out = empty_nhwc((n, y.shape[1], pp, pq), x)
    with H.guard(x.device):
        rc = H.lib().cpm_maxpool3x3s2_forward(H.ptr(y), n, p, q, y.shape[1], pp, pq, H.ptr(out), H.stream())
    H.check(rc, "maxpool")
    return out
