"""nn.Module containers whose forward runs the HIP kernels.

They subclass the torch modules only to inherit parameter shapes / names (the state-dict keys are part of the
reference's ABI) and initialisers; torch's own ATen kernels are never called on the hot path.
Conv weights are expected in channels_last memory (model.to(memory_format=torch.channels_last))."""
import torch
import torch.nn as nn

from . import conv as F


class Conv2d(nn.Conv2d):
    def forward(self, x, scale=None, shift=None, residual=None, relu=False, res_mode=0, sole_consumer=False,
                gate_by_consumers=False):
        assert self.padding[0] == self.padding[1] and self.stride[0] == self.stride[1] and self.padding_mode == "zeros"
        if shift is None:
            shift = self.bias
        else:
            assert self.bias is None
        if self.groups > 1 and self.in_channels // self.groups < 32 and self.kernel_size != (1, 1) \
                and residual is None:
            # narrow groups (ResNeXt 64x4d: 4..16 channels per group): sample columns once and contract R*S*C/g
            # deep instead of zero-padding every tap of every group to a 32-deep MFMA k-step
            from .deform_conv import cols_conv
            return cols_conv(x, None, self.weight, scale, shift, self.stride, self.padding, self.dilation,
                             self.groups, 1, relu, sole_consumer)
        return F.conv2d(x, self.weight, scale, shift, residual, self.stride[0], self.padding[0], self.dilation[0],
                        self.groups, relu, res_mode, sole_consumer, gate_by_consumers)


class Linear(nn.Linear):
    """nn.Linear; with `window=(C, H, W)` a Linear over a flattened [C,H,W] feature map (fc6 of the 2-FC heads,
    iou_fc1) whose weight is STORED as the full-window conv filter it is used as: logical [K,C,H,W], KRSC memory.
    The NHWC RoI features then need no repack, the weight no per-step permute, and the weight-gradient kernel
    accumulates straight into the flat gradient buffer.  The state dict keeps the reference's [K, C*H*W] tensor
    (column order c,h,w): converted on save / load, so checkpoints are unchanged."""

    def __init__(self, in_features, out_features, bias=True, window=None):
        super().__init__(in_features, out_features, bias)
        self.window = None
        if window is not None:
            c, h, w = (int(v) for v in window)
            assert c * h * w == in_features, "window does not match in_features"
            self.window = (c, h, w)
            w4 = self.weight.data.view(out_features, c, h, w).contiguous(memory_format=torch.channels_last)
            self.weight = nn.Parameter(w4)
            self.weight._cpm_abi_shape = (out_features, in_features)     # its shape in checkpoints (optimizer state too)
            self._register_state_dict_hook(Linear._flatten_on_save)
            self._register_load_state_dict_pre_hook(self._unflatten_on_load)

    @staticmethod
    def _flatten_on_save(module, state_dict, prefix, local_metadata):
        key = prefix + "weight"
        if key in state_dict:
            state_dict[key] = state_dict[key].reshape(module.out_features, module.in_features)

    def _unflatten_on_load(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                           error_msgs):
        key = prefix + "weight"
        v = state_dict.get(key)
        if v is not None and v.dim() == 2 and v.shape[1] == self.in_features:
            state_dict[key] = v.reshape((v.shape[0],) + self.window)

    def forward(self, x, relu=False, sole_consumer=False):
        """sole_consumer: the caller promises that the (ReLU) result feeds exactly one further Linear / conv of this
        package, whose data-gradient epilogue then applies this layer's ReLU gate (F.conv2d)."""
        if x.dim() == 4:
            # flatten of an NHWC feature map: run as a full-window conv so no NCHW repack of x is needed
            n, c, h, w = x.shape
            if self.window is not None:
                assert (c, h, w) == self.window, "feature map does not match the Linear's window"
                w4 = self.weight
            else:
                w4 = self.weight.view(self.out_features, c, h, w)   # reference column order c,h,w; repacked per call
            y = F.conv2d(x, w4, None, self.bias, None, 1, 0, 1, 1, relu, 0, sole_consumer)
            return F.carry_tag(y, y.reshape(n, self.out_features))
        w2 = self.weight if self.window is None else self.weight.reshape(self.out_features, self.in_features)
        return F.linear(x, w2, self.bias, relu, sole_consumer)


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x, relu=False):
        assert self.output_padding == (0, 0) and self.dilation == (1, 1)
        return F.conv_transpose2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.groups, relu)


class GroupNorm(nn.GroupNorm):
    def forward(self, x, relu=False):
        return F.group_norm(x, self.weight, self.bias, self.num_groups, self.eps, relu)


class ReLU(nn.ReLU):
    """Placeholder kept in nn.Sequential containers for key/index parity; the ReLU itself is fused upstream."""
