"""Detection ResNet body (counterpart of pet/rcnn/modeling/backbone/ResNet.py:24-148,288-301).

C2..C5 of a bottleneck ResNet with frozen affine norms; conv1 + layer1 frozen (FREEZE_AT=2).  The 7x7/s2 stem on
3 channels runs as im2col + one MFMA GEMM with affine+ReLU fused, followed by the 3x3/s2 max-pool kernel."""
import math

import torch
import torch.nn as nn

import pet.lib.ops as ops
import pet.models.imagenet.resnet as res
from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry
from pet.utils.net import freeze_params, make_norm


def get_norm():
    if cfg.BACKBONE.RESNET.USE_GN or cfg.BACKBONE.RESNET.USE_AN:
        raise ValueError("GN / mixture-norm backbones are outside the hot path")
    return "bn"


class ResNet(res.ResNet):
    def __init__(self, norm="bn", stride=32):
        super().__init__()
        R = cfg.BACKBONE.RESNET
        if not R.BOTTLENECK or R.USE_ALIGN or R.USE_3x3x3HEAD or R.USE_WS:
            raise ValueError("only the plain bottleneck ResNet is on the hot path")
        block = res.Bottleneck
        self.expansion = block.expansion
        self.stride_3x3, self.avg_down, self.norm, self.stride = R.STRIDE_3X3, R.AVG_DOWN, norm, stride
        layers = R.LAYERS[:int(math.log(stride, 2)) - 1]
        self.layers = layers
        self.base_width, self.ctx_ratio = R.WIDTH, R.CTX_RATIO
        swc, swx = R.STAGE_WITH_CONV, R.STAGE_WITH_CONTEXT
        self.inplanes = 64
        self.conv1 = ops.Conv2d(3, self.inplanes, 7, 2, 3, bias=False)
        self.bn1 = make_norm(self.inplanes, norm=self.norm.split("_")[-1])
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0], 1, conv=swc[0], context=swx[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2, conv=swc[1], context=swx[1])
        self.layer3 = self._make_layer(block, 256, layers[2], 2, conv=swc[2], context=swx[2])
        if len(layers) == 4:
            c5_stride = 1 if R.C5_DILATION != 1 else 2
            self.layer4 = self._make_layer(block, 512, layers[3], c5_stride, dilation=R.C5_DILATION, conv=swc[3],
                                           context=swx[3])
            self.spatial_scale = [1 / 4., 1 / 8., 1 / 16., 1 / 32. * R.C5_DILATION]
        else:
            self.spatial_scale = [1 / 4., 1 / 8., 1 / 16.]
        self.dim_out = self.stage_out_dim[1:int(math.log(self.stride, 2))]
        self._init_weights()
        self._init_modules()
        if cfg.MODEL.FPN_ON:
            # C2..C5 are consumed by the next stage's first block and by the FPN laterals: convolutions of this package
            # only, so the stage outputs' ReLU gates are applied by their consumers too (Bottleneck.gate_out)
            for li in range(1, len(layers) + 1):
                getattr(self, "layer%d" % li)[-1].gate_out = True
        self._stem_cache = None

    def _init_modules(self):
        fa = cfg.BACKBONE.RESNET.FREEZE_AT
        assert fa in [0, 2, 3, 4, 5] and fa <= len(self.layers) + 1
        if fa > 0:
            freeze_params(self.conv1)
            freeze_params(self.bn1)
        for i in range(1, fa):
            freeze_params(getattr(self, "layer%d" % i))
        self.apply(lambda m: freeze_params(m) if isinstance(m, ops.AffineChannel2d) else None)

    def _stem_weight(self):
        """conv1.weight [64,3,7,7] -> [64, 160, 1, 1]: (r,s,c) column order of cpm_im2col, zero padded."""
        w = self.conv1.weight
        key = (w.data_ptr(), w._version, str(w.device))
        if self._stem_cache is None or self._stem_cache[0] != key:
            k = w.shape[0]
            cols = w.shape[1] * w.shape[2] * w.shape[3]
            wp = torch.zeros((k, (cols + 31) // 32 * 32), dtype=w.dtype, device=w.device)
            wp[:, :cols] = w.detach().permute(0, 2, 3, 1).reshape(k, cols)
            self._stem_cache = (key, wp.view(k, -1, 1, 1))
        return self._stem_cache[1]

    def forward(self, x):
        if self.conv1.weight.requires_grad:
            raise RuntimeError("the stem runs forward-only (BACKBONE.RESNET.FREEZE_AT must be >= 2)")
        s, b = res._affine(self.bn1)
        x = ops.stem_forward(x, self._stem_weight(), s, b, 7, 7, 2, 3, w=self.conv1.weight)
        x2 = self.layer1(x)
        x3 = self.layer2(x2)
        x4 = self.layer3(x3)
        if len(self.layers) == 4:
            return [x2, x3, x4, self.layer4(x4)]
        return [x2, x3, x4]


@registry.BACKBONES.register("resnet")
def resnet():
    return ResNet(norm=get_norm())


@registry.BACKBONES.register("resnet_c4")
def resnet_c4():
    return ResNet(norm=get_norm(), stride=16)
