"""Split-plane ("SP") operands of the split-bf16 conv arithmetic (include/cpmrcnn_hip.h: cpm_split_planes,
cpm_conv2d_*_sp; kernel cpm-r-cnn_amd/csrc/conv_sp.hip).

An SP twin of an fp32 tensor holds, per memory row of C channels (an NHWC pixel, a (k, r, s) row of a KRSC weight;
C % 32 == 0) and per block of 32 channels, 32 bf16 `hi = bf16(v)` values followed by 32 bf16 `lo = bf16(v - hi)` values:
128 bytes = one cache line per (row, 32-channel reduction step), the same 4*C bytes as the fp32 row.  It
is carried as an int32 tensor of the fp32 tensor's shape and strides (opaque to torch: only the kernels read it) in the
attribute `_cpm_sp` of the tensor it mirrors; producers that can write it for free (conv epilogues) attach it, consumers
without one make it with one elementwise launch and cache it on the tensor for the other consumers (the weight
gradient reads the same activations and gradients as forward / data gradient)."""
import torch

from . import _hip as H

CL = torch.channels_last


def enabled():
    return H.get_conv_math() == "bf16x3"


def _rows_channels(t):
    """(rows, C) of the memory image: channels are the fastest-varying memory dimension."""
    if t.dim() == 4:
        n, c, h, w = t.shape
        if not t.is_contiguous(memory_format=CL):
            raise RuntimeError("SP operands are NHWC / KRSC in memory")
        return n * h * w, c
    if t.dim() == 2 and t.is_contiguous():
        return t.shape[0], t.shape[1]
    raise RuntimeError("SP operands are 4-D channels_last or 2-D contiguous tensors")


def split(t):
    """A fresh SP twin of fp32 tensor `t` (one launch); None when its channel count is not a multiple of 32 (twins
    are accelerators, never requirements: the convolutions then split in flight)."""
    rows, c = _rows_channels(t)
    if c % 32:
        return None
    sp = torch.empty_like(t, dtype=torch.int32)
    if t.numel():
        with H.guard(t.device):
            rc = H.lib().cpm_split_planes(H.ptr(t), H.c_int64(rows), int(c), H.ptr(sp), H.stream())
        H.check(rc, "split_planes")
    return sp


def of(t, make=True):
    """The SP twin cached on `t` (made and cached when absent and `make`); None when `t` cannot have one."""
    sp = getattr(t, "_cpm_sp", None)
    if sp is not None and sp._cpm_src_version == t._version:
        return sp
    if not make or t.dtype != torch.float32 or not t.is_cuda:
        return None
    try:
        rows, c = _rows_channels(t)
    except RuntimeError:
        return None
    if c % 32:
        return None
    sp = split(t)
    attach(t, sp)
    return sp


def attach(t, sp):
    """Record `sp` as the twin of `t` at its current version (an in-place edit of `t` invalidates it)."""
    sp._cpm_src_version = t._version
    t._cpm_sp = sp
    return t


def empty_like(t):
    return torch.empty_like(t, dtype=torch.int32)
