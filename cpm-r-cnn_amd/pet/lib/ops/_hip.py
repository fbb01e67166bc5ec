"""ctypes binding of libcpmrcnn_hip.so -- the only bridge between Python and the HIP kernels.

There is deliberately no CPU fallback: if the library is missing, or a tensor is not resident on
an MI355X, every op raises RuntimeError (the reference's AT_ASSERTM behaviour).
"""
import ctypes
import os
import threading

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
LIB_PATH = os.environ.get("CPM_LIB") or os.path.join(_PKG_ROOT, "lib", "libcpmrcnn_hip.so")   # CPM_LIB: A/B builds
_lib = None
_lock = threading.Lock()

c_int, c_float, c_void_p, c_size_t, c_int64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, \
    ctypes.c_int64


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "dilation", "groups", "P",
                                     "Q")]


class ChainLayer(ctypes.Structure):
    """cpm_chain_layer (include/cpmrcnn_hip.h)"""
    _fields_ = [("conv", ConvDesc)] + \
               [(n, c_void_p) for n in ("w", "wt", "bias", "gamma", "beta", "dw", "dbias", "dgamma", "dbeta")] + \
               [("has_gn", c_int), ("relu", c_int), ("gn_groups", c_int), ("eps", c_float), ("dgrad_flat", c_int),
                ("wt_w4", c_int), ("w4", c_void_p)]


def lib():
    """Load (once) and return the C-ABI library; raises if it was not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError("libcpmrcnn_hip.so not built: run `python cpm-r-cnn_amd/build.py` "
                                       "(or __graft_entry__.build()); there is no CPU fallback")
                # make sure the HIP runtime torch already loaded is the one we bind to
                tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
                if os.path.exists(tl):
                    ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
                L = ctypes.CDLL(LIB_PATH)
                L.cpm_last_error.restype = ctypes.c_char_p
                for name in ("cpm_nms_workspace_bytes", "cpm_conv2d_workspace_bytes",
                             "cpm_roi_align_fpn_gather_workspace_bytes", "cpm_sample_pos_neg_workspace_bytes"):
                    getattr(L, name).restype = c_size_t
                mode = os.environ.get("CPM_CONV_MATH", "").lower()
                if mode:
                    if mode not in CONV_MATH:
                        raise RuntimeError("CPM_CONV_MATH must be one of %s" % sorted(CONV_MATH))
                    L.cpm_set_conv_math(CONV_MATH[mode])
                _lib = L
    return _lib


CONV_MATH = {"f32": 0, "bf16x3": 1}


def set_conv_math(mode):
    """'f32' = exact fp32 MFMA; 'bf16x3' = 3-term split-bf16 MFMA with fp32 accumulation (cpmrcnn_hip.h)."""
    if mode not in CONV_MATH:
        raise RuntimeError("conv math must be one of %s" % sorted(CONV_MATH))
    rc = lib().cpm_set_conv_math(CONV_MATH[mode])
    check(rc, "set_conv_math")
    global _is_bf16x3
    _is_bf16x3 = mode == "bf16x3"


_is_bf16x3 = None


def bf16x3():
    """the conv arithmetic is bf16x3 (cached: asked once per conv call; only set_conv_math changes it)"""
    global _is_bf16x3
    if _is_bf16x3 is None:
        _is_bf16x3 = lib().cpm_get_conv_math() == CONV_MATH["bf16x3"]
    return _is_bf16x3


def get_conv_math():
    v = lib().cpm_get_conv_math()
    return [k for k, c in CONV_MATH.items() if c == v][0]


_deterministic = os.environ.get("CPM_DETERMINISTIC", "0") not in ("", "0")


def set_deterministic(on):
    """Split-K partial sums through ordered slab planes instead of float atomics (cpmrcnn_hip.h)."""
    global _deterministic
    check(lib().cpm_set_deterministic(int(bool(on))), "set_deterministic")
    _deterministic = bool(on)


def deterministic():
    """ordered reductions were asked for (set_deterministic / CPM_DETERMINISTIC): host-side choices between an
    atomic-scatter formulation and an ordered one follow it too"""
    return _deterministic


# Work a backward node has put off until a later point of the same backward pass (pooler_fpn: the heads' RoIAlign
# gradients, differentiated together).  Whoever is about to read what such work produces runs it first.
deferred = []


def run_deferred():
    while deferred:
        deferred.pop(0)()


_UNIT_SEEDS = {}


def unit_seed(device):
    """THE scalar 1 of this device: pet.utils.parallel.backward_losses seeds every loss term with it (no ones_like per
    term), and a loss node that receives it as its upstream gradient returns its stored gradient as is (scaled_by)."""
    one = _UNIT_SEEDS.get(device)
    if one is None:
        one = _UNIT_SEEDS[device] = torch.ones((), dtype=torch.float32, device=device)
    return one


def scaled_by(grad, g):
    """grad * g -- without the elementwise kernel when g IS the unit seed (a sum of several upstream gradients, or a
    weighted one, is a different tensor and is multiplied)."""
    one = _UNIT_SEEDS.get(g.device)
    if one is not None and g.dim() == 0 and g.data_ptr() == one.data_ptr():
        return grad
    return grad * g


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib().cpm_last_error().decode()))


# Host-side cost matters: the RoI-head phase of a training step is launch bound (~250 op calls between two host round
# trips), and `torch.cuda.current_stream()` / `torch.cuda.device()` cost ~8 us / ~5 us of Python per call.  The raw
# accessors below cost ~0.3 us.
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


_override = None        # raw hipStream_t the ops launch on instead of torch's current stream (see use_stream)
_override_ts = None     # its torch.cuda.Stream object when the section ALLOCATES on that stream's behalf (side_alloc)


def stream_raw():
    return _override if _override is not None else _raw_stream(_cur_device())


def stream():
    """The stream the ops launch on, as a raw hipStream_t: torch's current stream of the current device, or the one
    set by use_stream()."""
    return c_void_p(stream_raw())


class use_stream(object):
    """`with use_stream(raw):` -- every op launched inside goes to that raw stream (cheaper than torch.cuda.stream,
    and invisible to torch: tensors allocated inside still belong to torch's current stream).  `alloc_on`: the torch
    stream object of `raw` -- tensors the ops create inside are then allocated from THAT stream's pool (side_alloc)."""
    __slots__ = ("raw", "ts", "prev")

    def __init__(self, raw, alloc_on=None):
        self.raw, self.ts = raw, alloc_on

    def __enter__(self):
        global _override, _override_ts
        self.prev = (_override, _override_ts)
        _override, _override_ts = self.raw, self.ts

    def __exit__(self, *exc):
        global _override, _override_ts
        _override, _override_ts = self.prev
        return False


def side_alloc(fn):
    """Run the allocation(s) `fn()` makes for an op.  Inside a forward side section (conv.fwd_side) the op's kernel runs
    on the second stream, forked from the compute stream some way back: a block the caching allocator hands out from
    the COMPUTE stream's pool may have been freed by a tensor that compute-stream kernels queued behind the fork point
    are still reading (a bottleneck's conv1 output, dropped when conv2 returned, became its downsample conv's output:
    the X-101 body read garbage whenever the pool offered that block).  So such tensors come from the second stream's
    own pool, and the compute stream is recorded as a user (their consumers run there behind the join), which makes
    the allocator wait for it before the block is handed on."""
    if _override is None or _override_ts is None or torch.cuda.is_current_stream_capturing():
        return fn()             # (under a hipGraph capture every allocation belongs to the capture's private pool)
    main = torch.cuda.current_stream(_override_ts.device)
    with torch.cuda.stream(_override_ts):
        out = fn()
    for t in (out if isinstance(out, (tuple, list)) else (out,)):
        if t is not None:
            t.record_stream(main)
    return out


# Tensors created inside a use_stream() section are compute-stream blocks to torch's caching allocator (torch's current
# stream never changes), while the kernels that write them are queued on the override stream.  One that is dropped
# before the compute stream has been made to wait for that stream would go back to the compute-stream pool with the side
# kernel still pending, and the next compute-stream allocation could receive it.  keep() parks such tensors (RoIAlign's
# level indices, layout / contiguity copies); release_kept() -- called by the joins, conv.fwd_join / conv._join_side --
# drops them once the compute stream waits for the side stream.
_side_keep = []


def in_side_section():
    return _override is not None


def keep(*tensors):
    """inside a use_stream() section: hold these tensors until the next join of the two streams"""
    if _override is not None:
        _side_keep.extend(t for t in tensors if t is not None)


def release_kept():
    del _side_keep[:]


def fork(raw_from, raw_to):
    """stream `raw_to` waits for everything queued on `raw_from` so far (cpm_stream_fork)"""
    check(lib().cpm_stream_fork(c_void_p(raw_from), c_void_p(raw_to)), "stream_fork")


class _NoGuard(object):
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def guard(device):
    """Context that makes `device` current -- a no-op object when it already is (one process per GPU: always)."""
    idx = device.index
    if idx is None or idx == _cur_device():
        return _NO_GUARD
    return torch.cuda.device(device)


# ---- the optimizer step beside the next forward pass (FlatSGD.overlap_next_forward) -----------------------------------
# The SGD kernel (0.8 ms of pure HBM streaming for R-50) and the once-per-step weight-image transform run on a stream of
# their own; the next step's FROZEN layers (stem, layer1: ~1 ms that read no trainable tensor) start beside them, and the
# compute stream waits for the update at the first op that takes a tensor of the flat parameter buffer (require_gpu
# below sees every op's operands) -- or wherever else parameters are read (wait_pending_sgd: zero_grad's memset, state
# dicts, graph replays).
_pending_sgd = {}        # device index -> [events]
_opt_streams = {}
_aux_streams = {}        # device index -> [torch streams this package launches kernels on besides the compute stream]


def register_aux_stream(device, stream):
    """a stream kernels of this package may be queued on beside torch's current one (the second / weight-gradient
    stream, the deformable parameter-gradient stream): whoever orders "everything that reads parameters" behind an
    event -- wait_pending_sgd -- orders these too.  (A side section forked BEFORE the first trainable op of a forward
    pass would otherwise run its trainable conv -- a stage's downsample -- without the wait the compute stream made.)"""
    idx = device.index if device.index is not None else _cur_device()
    lst = _aux_streams.setdefault(idx, [])
    if all(s is not stream for s in lst):
        lst.append(stream)


def optimizer_stream(device):
    idx = device.index if device.index is not None else _cur_device()
    st = _opt_streams.get(idx)
    if st is None:
        st = _opt_streams[idx] = torch.cuda.Stream(device=device)
    return st


def set_pending_sgd_event(ev, device):
    idx = device.index if device.index is not None else _cur_device()
    _pending_sgd.setdefault(idx, []).append(ev)


def wait_pending_sgd(device=None):
    """the current stream of `device` (default: every device with an update outstanding) waits for the optimizer
    updates queued on the optimizer stream"""
    if not _pending_sgd:
        return
    idxs = list(_pending_sgd) if device is None else [device.index if device.index is not None else _cur_device()]
    for idx in idxs:
        evs = _pending_sgd.pop(idx, None)
        if evs:
            for st in [torch.cuda.current_stream(idx)] + _aux_streams.get(idx, []):
                for ev in evs:
                    st.wait_event(ev)


def require_gpu(*tensors):
    if _pending_sgd:
        for t in tensors:
            if t is not None and (getattr(t, "_cpm_grad_sink", None) is not None or getattr(t, "_cpm_owner", None) is not None):
                wait_pending_sgd(t.device)
                break
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("cpm-r-cnn_amd ops run on MI355X only: got a %s tensor (no CPU fallback)" % t.device)
        if t.dtype not in (torch.float32, torch.int64, torch.int32):
            raise RuntimeError("unsupported dtype %s (ops are fp32-only, like the reference's amp.float_function)"
                               % t.dtype)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def f(x):
    return c_float(float(x))


_ws = {}
_ws_retired = []        # buffers replaced while a non-compute stream may still use them (see workspace / release_retired)


def workspace(nbytes, device):
    """A grow-only scratch buffer per (device, stream); callers never hold it across ops.

    Under use_stream() the buffer belongs to the override (side) stream, but torch's caching allocator only knows the
    compute stream it was allocated on: dropping it on growth would let the allocator hand the block straight back to
    the compute stream while side-stream kernels already queued (weight-gradient slab planes in deterministic mode)
    still use it.  A replaced side-stream buffer is therefore parked until release_retired() -- called once the compute
    stream has been made to wait for the side stream (conv._join_side) -- and it grows geometrically so that a rising
    RoI count parks O(log) buffers."""
    idx = device.index if device.index is not None else _cur_device()
    key = (idx, _override if _override is not None else _raw_stream(idx))
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        want = max(int(nbytes), 1 << 20)
        if buf is not None and _override is not None:
            _ws_retired.append(buf)
            want = max(want, 2 * buf.numel())
        buf = side_alloc(lambda: torch.empty(want, dtype=torch.uint8, device=device))
        _ws[key] = buf
    return buf


def release_retired():
    """Drop the parked side-stream workspaces: only after the compute stream waits for everything the side stream has
    been given (the join at the end of a backward pass)."""
    del _ws_retired[:]
