"""nn.Module containers whose forward runs the HIP kernels.

They subclass the torch modules only to inherit parameter shapes / names (the state-dict keys are part of the
reference's ABI) and initialisers; torch's own ATen kernels are never called on the hot path.
Conv weights are expected in channels_last memory (model.to(memory_format=torch.channels_last))."""
import torch
import torch.nn as nn

from . import conv as F


class Conv2d(nn.Conv2d):
    def forward(self, x, scale=None, shift=None, residual=None, relu=False, res_mode=0, sole_consumer=False):
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
                        self.groups, relu, res_mode, sole_consumer)


class Linear(nn.Linear):
    def forward(self, x, relu=False):
        if x.dim() == 4:
            # flatten of an NHWC feature map: run as a full-window conv so no NCHW repack of x is needed;
            # the [K, C*H*W] weight (reference column order c,h,w) is viewed as [K,C,H,W]
            n, c, h, w = x.shape
            w4 = self.weight.view(self.out_features, c, h, w)
            y = F.conv2d(x, w4, None, self.bias, None, 1, 0, 1, 1, relu, 0)
            return y.reshape(n, self.out_features)
        return F.linear(x, self.weight, self.bias, relu)


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x, relu=False):
        assert self.output_padding == (0, 0) and self.dilation == (1, 1)
        return F.conv_transpose2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.groups, relu)


class GroupNorm(nn.GroupNorm):
    def forward(self, x, relu=False):
        return F.group_norm(x, self.weight, self.bias, self.num_groups, self.eps, relu)


class ReLU(nn.ReLU):
    """Placeholder kept in nn.Sequential containers for key/index parity; the ReLU itself is fused upstream."""
