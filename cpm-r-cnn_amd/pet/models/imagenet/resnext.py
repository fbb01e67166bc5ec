"""ResNeXt bottleneck body (counterpart of pet/models/imagenet/resnext.py:14-83,173-297).

Block layout and parameter names are the reference's (conv1/bn1 1x1, conv2/bn2 grouped 3x3 carrying the stride --
an `ops.DeformConvPack` with its `conv_offset` child when the stage is 'deform' --, conv3/bn3 1x1,
downsample.0/.1), so `resnext101b_64x4d` checkpoints map key-for-key.  Every conv runs on the HIP kernels with the
frozen affine, residual add and ReLU fused; the grouped 3x3 (4..32 channels per group) goes through the column
path of pet/lib/ops/deform_conv.py."""
import math

import torch.nn as nn

import pet.lib.ops as ops
from pet.models.imagenet.resnet import _affine
from pet.utils.net import make_norm


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, base_width, cardinality, stride=1, dilation=1, norm="bn", conv="normal",
                 context="none", ctx_ratio=0.0625, downsample=None):
        super().__init__()
        D = int(math.floor(planes * (base_width / 64.0)))
        C = cardinality
        if conv == "normal":
            conv_op = ops.Conv2d
        elif conv == "deform":
            conv_op = ops.DeformConvPack
        else:
            raise ValueError("{} type conv operation is not on the hot path (dcn v1 and normal are).".format(conv))
        if context != "none":
            raise ValueError("{} type context operation is outside the hot path.".format(context))
        self.conv1 = ops.Conv2d(inplanes, D * C, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn1 = make_norm(D * C, norm=norm)
        self.conv2 = conv_op(D * C, D * C, kernel_size=3, stride=stride, dilation=dilation, padding=dilation,
                             groups=C, bias=False)
        self.bn2 = make_norm(D * C, norm=norm)
        self.conv3 = ops.Conv2d(D * C, planes * 4, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn3 = make_norm(planes * 4, norm=norm)
        self.ctx = None
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        # every consumer of this block's output is a convolution of pet.lib.ops (or the residual add of one): they apply
        # its ReLU gate inside their data-gradient kernels (ops.conv2d: gate_by_consumers).  Set by whoever wires the
        # blocks together (_make_layer: blocks inside a stage; the detection backbone: stage outputs under an FPN)
        self.gate_out = False

    def forward(self, x):
        ops.mark_shared_grad(x)       # consumers: conv1, the downsample conv or conv3's residual -- all in-package
        # the downsample conv reads x only: its kernel runs on the second stream beside conv1 -> conv2 (ops.fwd_fork)
        forked = self.downsample is not None and ops.fwd_fork(x, 1)
        s, b = _affine(self.bn1)
        # conv1's output feeds conv2 alone, or -- a DeformConvPack -- its sampled conv and its offset predictor, whose
        # data-gradient kernel (the last of the two to run) applies conv1's ReLU gate to their summed gradient
        out = self.conv1(x, scale=s, shift=b, relu=True, gate_by_consumers=isinstance(self.conv2, ops.DeformConvPack))
        s, b = _affine(self.bn2)
        out = self.conv2(out, scale=s, shift=b, relu=True, sole_consumer=True)     # consumed by conv3 only
        residual = x
        if self.downsample is not None:
            s, b = _affine(self.downsample[1])
            if forked:
                with ops.fwd_side(x):
                    residual = self.downsample[0](x, scale=s, shift=b)
                ops.fwd_join(x)
            else:
                residual = self.downsample[0](x, scale=s, shift=b)
        s, b = _affine(self.bn3)
        return self.conv3(out, scale=s, shift=b, residual=residual, relu=True, gate_by_consumers=self.gate_out)


class ResNeXt(nn.Module):
    """Parameter container + layer builder shared with the detection backbone (resnext.py:173-297)."""

    def __init__(self):
        super().__init__()

    @property
    def stage_out_dim(self):
        return [64, 64 * self.expansion, 128 * self.expansion, 256 * self.expansion, 512 * self.expansion]

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.0001)
                nn.init.constant_(m.bias, 0)
        for m in self.modules():                       # zero-init the offset predictors (resnext.py:248-252)
            if isinstance(m, ops.DeformConvPack):
                nn.init.constant_(m.conv_offset.weight, 0)
                nn.init.constant_(m.conv_offset.bias, 0)
        for m in self.modules():                       # zero gamma of each block's last norm (:256-259)
            if isinstance(m, Bottleneck):
                nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, conv="normal", context="none"):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            if self.avg_down:
                raise ValueError("AVG_DOWN is outside the hot path")
            downsample = nn.Sequential(
                ops.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                make_norm(planes * block.expansion, norm=self.norm))
        layers = [block(self.inplanes, planes, self.base_width, self.cardinality, stride, dilation, self.norm, conv,
                        context, self.ctx_ratio, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, self.base_width, self.cardinality, 1, dilation, self.norm,
                                conv, context, self.ctx_ratio))
        for blk in layers[:-1]:            # consumed by the next block's conv1 / residual add only
            blk.gate_out = True
        return nn.Sequential(*layers)
