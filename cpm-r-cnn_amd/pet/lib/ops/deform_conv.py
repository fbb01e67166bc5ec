"""Deformable convolution v1 and narrow-group convolution (counterpart of pet/lib/ops/deform_conv.py:13-186,
323-400,472-514; native side pet/lib/ops/csrc/Deformable/deform_conv_cuda.cu).

Same call surface as the reference (`deform_conv(input, offset, weight, bias, stride, padding, dilation, groups,
deformable_groups)`, `DeformConv`, `DeformConvPack` with its zero-initialised `conv_offset`), different engine:
the sampled columns are written pixel-major (`cpm_deform_im2col`) so that the contraction, its data gradient and
its weight gradient are ONE grouped 1x1 problem each on the MFMA implicit-GEMM kernels, with the frozen affine /
bias / ReLU fused in the epilogue; `cpm_deform_col2im` / `cpm_deform_coord_grad` scatter the column gradient back
to the input and the offsets.  `offset=None` runs the same path as a plain im2col: that is how ResNeXt's ordinary
3x3 convs with 4..32 channels per group avoid zero-padding every group to a 32-deep MFMA k-step."""
import os

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import _hip as H
from . import conv as F
from .modules import Conv2d


def _geom(x_shape, w_shape, stride, pad, dil, groups, dg):
    n, c, h, w = x_shape
    k, cg, r, s = w_shape
    if cg * groups != c:
        raise RuntimeError("weight [%d,%d,%d,%d] does not match %d input channels in %d groups" % (k, cg, r, s, c,
                                                                                                    groups))
    p, q = F.out_size(h, r, stride, pad, dil), F.out_size(w, s, stride, pad, dil)
    return (int(n), int(h), int(w), int(c), int(r), int(s), int(stride), int(pad), int(dil), int(groups), int(dg),
            int(p), int(q))


def _check_offset(offset, geom):
    n, _, _, _, r, s, _, _, _, _, dg, p, q = geom
    if tuple(offset.shape) != (n, 2 * r * s * dg, p, q):
        raise RuntimeError("offset must be [%d, %d, %d, %d], got %s" % (n, 2 * r * s * dg, p, q,
                                                                        tuple(offset.shape)))


def sample_columns(x, offset, geom):
    """cols [N*P*Q, groups*R*S*C/groups]; logical NCHW view [N, R*S*C, P, Q] in NHWC memory."""
    n, h, w, c, r, s, stride, pad, dil, groups, dg, p, q = geom
    cols = torch.empty((n, p, q, r * s * c), dtype=torch.float32, device=x.device)
    if cols.numel():
        with H.guard(x.device):
            rc = H.lib().cpm_deform_im2col(H.ptr(x), H.ptr(offset), n, h, w, c, r, s, stride, pad, dil, groups, dg,
                                           p, q, H.ptr(cols), H.stream())
        H.check(rc, "deform_im2col")
    return cols.permute(0, 3, 1, 2)


_fused = [True]
# the fused backward is two kernels that do not depend on each other -- the data gradient (dx through an LDS window)
# and the parameter gradients (dw, d offset from the x window) -- each of which holds a CU's LDS with two workgroups of
# four waves: they run side by side, the second on a stream of its own, joined before the backward returns
_OVERLAP = os.environ.get("CPM_DEFORM_OVERLAP", "1") != "0"
_par_streams = {}


def _par_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _par_streams.get(idx)
    if st is None:
        t = torch.cuda.Stream(device=device)
        st = _par_streams[idx] = (t, t.cuda_stream)
        H.register_aux_stream(device, t)
    return st


_capture = None          # tools/deform_capture.py: a list that receives the inputs of every fused backward


def set_fused(on):
    """tests / A-B runs: route every layer through the column-matrix path (False) or let the fused kernels take the
    layers they support (True, the default; CPM_DEFORM_FUSED=0 in the environment switches them off in the library)"""
    prev = _fused[0]
    _fused[0] = bool(on)
    return prev


def fused_ok(geom, k):
    """the sampling-inside-the-contraction kernels (csrc/deform_fused.hip) take this layer"""
    n, h, w, c, r, s, stride, pad, dil, groups, dg, p, q = geom
    # deterministic mode (cpm_set_deterministic / CPM_DETERMINISTIC: bit-identical weight gradients, cpmrcnn_hip.h): the
    # fused parameter-gradient kernel adds its per-workgroup dw blocks with float atomics, so the layer takes the column
    # path, whose weight gradient is conv2d_backward_weight's ordered slab reduction
    if H.deterministic():
        return False
    return _fused[0] and bool(n and p and q) and bool(H.lib().cpm_deform_conv_fused_supported(n, h, w, c, int(k), r, s, stride, pad,
                                                                                dil, groups, dg, p, q))


def _fused_args(geom, k):
    n, h, w, c, r, s, stride, pad, dil, groups, dg, p, q = geom
    return (n, h, w, c, int(k), r, s, stride, pad, dil, groups, dg, p, q)


def _w1x1(w):
    """[K, C/g, R, S] in KRSC memory -> the same bytes as a [K, R*S*C/g, 1, 1] weight (no copy)."""
    k, cg, r, s = w.shape
    return w.permute(0, 2, 3, 1).reshape(k, r * s * cg, 1, 1)


def _param_grads(ctx, need_w, params, cols, dpre, w, w1, doff, r, s, groups, dy):
    """dw (in the parameter's sink, or returned) and -- fused -- d offset by the same kernel"""
    dw = done_wp = None
    if need_w:
        wp = ctx.wparam
        if wp is not None and wp._cpm_grad_sink.data_ptr() != 0 and w.data_ptr() == wp.data_ptr():
            if ctx.fused:
                params(wp._cpm_grad_sink, doff)
            else:
                F.conv2d_backward_weight(cols, dpre, w1, 1, 0, 1, groups, out=wp._cpm_grad_sink)
            done_wp = wp                            # announced at the end: the data gradient still reads w
        else:
            if wp is not None:                      # this use reaches the parameter through autograd's accumulation
                wp._cpm_uses -= 1
            k, cg = w.shape[0], w.shape[1]
            if ctx.fused:
                dw1 = torch.zeros((k, r * s * cg, 1, 1), dtype=torch.float32, device=dy.device)
                params(dw1, doff)
            else:
                dw1 = F.conv2d_backward_weight(cols, dpre, w1, 1, 0, 1, groups)
            dw = dw1.view(k, r, s, cg).permute(0, 3, 1, 2)
    elif doff is not None:
        params(None, doff)
    return dw, done_wp


class _ColsConvFn(Function):
    """y = relu?( contract(columns(x, offset), w) * scale + shift )"""

    @staticmethod
    def forward(ctx, x, offset, w, scale, shift, stride, pad, dil, groups, dg, relu, out_tag=None):
        H.require_gpu(x, offset, w, scale, shift)
        # the input may be a multi-consumer activation with an in-place gradient accumulator (conv.mark_shared_grad:
        # DeformConvPack's input feeds this node and the offset predictor) and carry its producer's ReLU-gate tag
        ctx.x_holder = getattr(x, "_cpm_gacc", None)
        ctx.in_tag = getattr(x, "_cpm_epi", None)
        x = F.nhwc(x)
        w_in = w
        w = F._wmem(w)
        geom = _geom(x.shape, w.shape, stride, pad, dil, groups, dg)
        if offset is not None:
            offset = F.nhwc(offset)
            _check_offset(offset, geom)
        ctx.wparam = w_in if (ctx.needs_input_grad[2] and getattr(w_in, "_cpm_grad_sink", None) is not None) else None
        if ctx.wparam is not None:
            F._note_use(w_in)
        ctx.fused = fused_ok(geom, w.shape[0])
        if ctx.fused:
            # sampling inside the contraction, exact-f32 MFMA in either conv arithmetic: no column matrix
            cols = None
            y = F.empty_nhwc((geom[0], w.shape[0], geom[11], geom[12]), x)
            with H.guard(x.device):
                rc = H.lib().cpm_deform_conv_forward(H.ptr(x), H.ptr(offset), H.ptr(w), H.ptr(scale), H.ptr(shift),
                                                     int(bool(relu)), *_fused_args(geom, w.shape[0]), H.ptr(y),
                                                     H.stream())
            H.check(rc, "deform_conv_forward")
        else:
            cols = sample_columns(x, offset, geom)
            # the parameter's pre-split image (bf16x3) is the image of the [K, R*S*C/g, 1, 1] weight too: the same bytes
            y = F.conv2d_forward(cols, _w1x1(w), scale, shift, None, 0, relu, 1, 0, 1, groups,
                                 w4=F.w4_of(w_in, w, (w.shape[1] * w.shape[2] * w.shape[3])))
        ctx.geom, ctx.relu = geom, relu
        ctx.out_tag = out_tag                      # see conv._ConvFn: gate + scale applied by the sole consumer's dgrad
        ctx.has = (scale is not None, shift is not None)
        need_cols = ctx.needs_input_grad[2] and not ctx.fused
        ctx.save_for_backward(x, offset, w, scale, y if relu else None, cols if need_cols else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, offset, w, scale, y, cols = ctx.saved_tensors
        n, h, wd, c, r, s, stride, pad, dil, groups, dg, p, q = ctx.geom
        has_scale, has_shift = ctx.has
        need_x, need_off, need_w, _, need_shift = ctx.needs_input_grad[:5]
        dy = F.nhwc(dy)
        dpre, dshift = dy, None
        gated = ctx.out_tag is not None and ctx.out_tag["applied"]
        if not gated and (ctx.relu or has_scale or (has_shift and need_shift)):
            dpre_k, _, dshift = F.epilogue_backward(dy, y, scale, ctx.relu, want_dpre=ctx.relu or has_scale,
                                                    want_dshift=has_shift and need_shift)
            dpre = dpre_k if dpre_k is not None else dy
        w1 = _w1x1(w)
        dw = None
        done_wp = None
        fargs = _fused_args(ctx.geom, w.shape[0])
        need_off = need_off and offset is not None
        if _capture is not None and ctx.fused:
            _capture.append((dpre.clone(), x, offset, w, fargs))

        def params(out, doff):
            """fused: dw (+)= and d offset in ONE kernel (both read the four corners of every sample)"""
            if dpre.numel() == 0:
                if doff is not None:
                    doff.zero_()
                return
            with H.guard(dy.device):
                rc = H.lib().cpm_deform_conv_backward_params(H.ptr(dpre), H.ptr(x), H.ptr(offset), H.ptr(w), *fargs,
                                                             H.ptr(out), H.ptr(doff), H.stream())
            H.check(rc, "deform_conv_backward_params")

        dx = doff = None
        if ctx.fused and need_off:
            doff = torch.empty_like(offset)
        side = None
        wp = ctx.wparam
        in_sink = wp is not None and wp._cpm_grad_sink.data_ptr() != 0 and w.data_ptr() == wp.data_ptr()
        # (a dw that is returned is zero-filled by torch on the compute stream: that route stays on it)
        if ctx.fused and _OVERLAP and need_x and (need_w or need_off) and (in_sink or not need_w) and dpre.numel():
            side = _par_stream(dy.device)[1]
            main_raw = H.stream_raw()
            H.fork(main_raw, side)                  # dpre, the sink's earlier writers
            params_on = H.use_stream(side)
        else:
            params_on = H._NoGuard()
        with params_on:
            dw, done_wp = _param_grads(ctx, need_w, params, cols, dpre, w, w1, doff, r, s, groups, dy)
        if ctx.fused:
            if need_x:
                # the tap's column gradient in registers -> dx through an LDS window
                dx = F.empty_nhwc((n, c, h, wd), dy).zero_()
                if dpre.numel():
                    with H.guard(dy.device):
                        rc = H.lib().cpm_deform_conv_backward_data(H.ptr(dpre), H.ptr(offset), H.ptr(w), *fargs,
                                                                   H.ptr(dx), H.stream())
                    H.check(rc, "deform_conv_backward_data")
            if side is not None:
                H.fork(side, main_raw)              # the compute stream waits for dw / d offset
        elif need_x or need_off:
            dcols = F.conv2d_backward_data(dpre, w1, (n, r * s * c, p, q), 1, 0, 1, groups)
            args = (n, h, wd, c, r, s, stride, pad, dil, groups, dg, p, q)
            with H.guard(dy.device):
                if need_x:
                    dx = F.empty_nhwc((n, c, h, wd), dy).zero_()
                    if dcols.numel():
                        rc = H.lib().cpm_deform_col2im(H.ptr(dcols), H.ptr(offset), *args, H.ptr(dx), H.stream())
                        H.check(rc, "deform_col2im")
                if need_off:
                    doff = torch.empty_like(offset)
                    if dcols.numel():
                        rc = H.lib().cpm_deform_coord_grad(H.ptr(dcols), H.ptr(x), H.ptr(offset), *args, H.ptr(doff),
                                                           H.stream())
                        H.check(rc, "deform_coord_grad")
        if dx is not None:
            if ctx.in_tag is not None:
                ctx.in_tag["applied"] = False       # a contribution without the producer's gate: a later consumer's
            h = ctx.x_holder                        # data-gradient kernel masks the running sum, or the producer does
            if h is not None:
                if "acc" in h and tuple(h["acc"].shape) == tuple(dx.shape):
                    F._wait_readers(h)
                    h["acc"].add_(dx)
                    dx = None
                else:
                    h["acc"] = dx                   # handed to autograd below; later consumers add into it in place
        if done_wp is not None:
            F._sink_done(done_wp)
        return dx, doff, dw, None, dshift, None, None, None, None, None, None, None


def _one(v):
    a, b = _pair(v)
    if a != b:
        raise RuntimeError("only square stride / padding / dilation are on the hot path")
    return int(a)


def cols_conv(x, offset, weight, scale=None, shift=None, stride=1, padding=0, dilation=1, groups=1,
              deformable_groups=1, relu=False, sole_consumer=False):
    tag = None
    if sole_consumer and relu and not (shift is not None and shift.requires_grad) and torch.is_grad_enabled():
        tag = {"scale": scale, "applied": False}
    y = _ColsConvFn.apply(x, offset, weight, scale, shift, _one(stride), _one(padding), _one(dilation),
                          int(groups), int(deformable_groups), bool(relu), tag)
    if tag is not None:
        y._cpm_epi = tag
    return y


def deform_conv(input, offset, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
    """Reference signature (deform_conv.py:13-27 + the bias add of :375-388); im2col_step is accepted and unused --
    the whole batch is one launch here."""
    if offset is None:
        raise RuntimeError("deform_conv needs an offset tensor")
    return cols_conv(input, offset, weight, None, bias, stride, padding, dilation, groups, deformable_groups)


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        self.with_bias = bias
        assert in_channels % groups == 0, \
            "in_channels {} cannot be divisible by groups {}".format(in_channels, groups)
        assert out_channels % groups == 0, \
            "out_channels {} cannot be divisible by groups {}".format(out_channels, groups)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)

    def _run(self, x, offset, scale, shift, relu, sole_consumer=False):
        if shift is None:
            shift = self.bias
        else:
            assert self.bias is None
        return cols_conv(x, offset, self.weight, scale, shift, self.stride, self.padding, self.dilation, self.groups,
                         self.deformable_groups, relu, sole_consumer)

    def forward(self, x, offset, scale=None, shift=None, relu=False, sole_consumer=False):
        return self._run(x, offset, scale, shift, relu, sole_consumer)

    def extra_repr(self):
        return ("in_channels={in_channels}, out_channels={out_channels}, kernel_size={kernel_size}, stride={stride}, "
                "padding={padding}, dilation={dilation}, groups={groups}, deformable_groups={deformable_groups}, "
                "bias={with_bias}").format(**self.__dict__)


class DeformConvPack(DeformConv):
    """DeformConv that predicts its own offsets with a zero-initialised conv (deform_conv.py:472-513)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         deformable_groups, bias)
        self.conv_offset = Conv2d(self.in_channels,
                                  self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                  kernel_size=self.kernel_size, stride=_pair(self.stride),
                                  padding=_pair(self.padding), bias=True)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x, scale=None, shift=None, relu=False, sole_consumer=False):
        # x has two consumers, both of this package: the offset predictor adds its data gradient INTO the tensor the
        # sampled conv handed to autograd (and applies the gate of x's producer to the sum when x carries its tag)
        # instead of autograd adding two tensors and the producer running a gate pass of its own
        F.mark_shared_grad(x)
        offset = self.conv_offset(x)
        return self._run(x, offset, scale, shift, relu, sole_consumer)
