"""ResNet bottleneck body (counterpart of pet/models/imagenet/resnet.py:71-136,218-339).

Module / parameter names equal the reference's (conv1,bn1,...,downsample.0,downsample.1) so ImageNet and
released checkpoints map key-for-key.  Every conv runs on the implicit-GEMM HIP kernel with the following
frozen affine, the residual add and the ReLU fused into its epilogue -- one launch per conv instead of the
reference's conv + mul + add + add + relu kernels (SURVEY 2b "fused ops: none")."""
import torch.nn as nn

import pet.lib.ops as ops
from pet.utils.net import make_norm


def _affine(norm):
    if not isinstance(norm, ops.AffineChannel2d):
        raise RuntimeError("the HIP backbone needs frozen affine norms: build with MODEL.BATCH_NORM='freeze' and "
                           "run convert_bn2affine_model (got %s)" % type(norm).__name__)
    return norm.weight, norm.bias


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, base_width=64, stride=1, dilation=1, norm="bn", conv="normal",
                 context="none", ctx_ratio=0.0625, stride_3x3=False, downsample=None):
        super().__init__()
        if conv != "normal" or context != "none":
            raise ValueError("deformable / context blocks are not built in this round (conv=%s, context=%s)"
                             % (conv, context))
        s1, s3 = (1, stride) if stride_3x3 else (stride, 1)
        width = int(planes * (base_width / 64.))
        self.conv1 = ops.Conv2d(inplanes, width, kernel_size=1, stride=s1, bias=False)
        self.bn1 = make_norm(width, norm=norm.split("_")[-1])
        self.conv2 = ops.Conv2d(width, width, kernel_size=3, stride=s3, dilation=dilation, padding=dilation,
                                bias=False)
        self.bn2 = make_norm(width, norm=norm)
        self.conv3 = ops.Conv2d(width, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = make_norm(planes * self.expansion, norm=norm.split("_")[-1])
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
        out = self.conv1(x, scale=s, shift=b, relu=True, sole_consumer=True)       # consumed by conv2 only
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


class ResNet(nn.Module):
    """Parameter container + layer builder shared with the detection backbone (resnet.py:218-339)."""

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
        for m in self.modules():                       # zero gamma of each block's last norm (resnet.py:299-306)
            if isinstance(m, Bottleneck):
                nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, conv="normal", context="none"):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            if self.avg_down:
                raise ValueError("AVG_DOWN is outside the hot path")
            downsample = nn.Sequential(
                ops.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                make_norm(planes * block.expansion, norm=self.norm.split("_")[-1]))
        layers = [block(self.inplanes, planes, self.base_width, stride, dilation, self.norm, conv, context,
                        self.ctx_ratio, self.stride_3x3, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, self.base_width, 1, dilation, self.norm, conv, context,
                                self.ctx_ratio, self.stride_3x3))
        for blk in layers[:-1]:            # consumed by the next block's conv1 / residual add only
            blk.gate_out = True
        return nn.Sequential(*layers)
