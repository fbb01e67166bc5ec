"""Detection ResNeXt body (counterpart of pet/rcnn/modeling/backbone/ResNeXt.py:19-141): C2..C5 of a grouped
bottleneck ResNeXt (BASELINE config #5: 64x4d, LAYERS (3,4,23,3), STAGE_WITH_CONV ('normal','deform','deform',
'deform')) with frozen affine norms and conv1 + layer1 frozen.  Stem as in backbone/ResNet.py here."""
import math

import torch.nn as nn

import pet.lib.ops as ops
import pet.models.imagenet.resnext as resx
from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry
from pet.rcnn.modeling.backbone.ResNet import ResNet as _ResNetBody
from pet.utils.net import freeze_params, make_norm


def get_norm():
    if cfg.BACKBONE.RESNEXT.USE_GN:
        raise ValueError("GN backbones are outside the hot path")
    return "bn"


class ResNeXt(resx.ResNeXt):
    def __init__(self, norm="bn", stride=32):
        super().__init__()
        R = cfg.BACKBONE.RESNEXT
        if R.USE_ALIGN or R.USE_3x3x3HEAD or R.USE_WS:
            raise ValueError("only the plain bottleneck ResNeXt is on the hot path")
        block = resx.Bottleneck
        self.expansion = block.expansion
        self.avg_down, self.norm, self.stride = R.AVG_DOWN, norm, stride
        self.cardinality, self.base_width, self.ctx_ratio = R.C, R.WIDTH, R.CTX_RATIO
        layers = R.LAYERS
        self.layers = layers
        swc, swx = R.STAGE_WITH_CONV, R.STAGE_WITH_CONTEXT
        self.inplanes = 64
        self.conv1 = ops.Conv2d(3, self.inplanes, 7, 2, 3, bias=False)
        self.bn1 = make_norm(self.inplanes, norm=self.norm)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0], 1, conv=swc[0], context=swx[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2, conv=swc[1], context=swx[1])
        self.layer3 = self._make_layer(block, 256, layers[2], 2, conv=swc[2], context=swx[2])
        self.layer4 = self._make_layer(block, 512, layers[3], 1 if R.C5_DILATION != 1 else 2,
                                       dilation=R.C5_DILATION, conv=swc[3], context=swx[3])
        self.spatial_scale = [1 / 4., 1 / 8., 1 / 16., 1 / 32. * R.C5_DILATION]
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
        fa = cfg.BACKBONE.RESNEXT.FREEZE_AT
        assert fa in [0, 2, 3, 4, 5] and fa <= len(self.layers) + 1
        if fa > 0:
            freeze_params(self.conv1)
            freeze_params(self.bn1)
        for i in range(1, fa):
            freeze_params(getattr(self, "layer%d" % i))
        self.apply(lambda m: freeze_params(m) if isinstance(m, ops.AffineChannel2d) else None)

    def train(self, mode=True):
        # frozen stages stay in eval mode (ResNeXt.py:91-105); with frozen affines this only sets flags
        self.training = mode
        for i in range(max(cfg.BACKBONE.RESNEXT.FREEZE_AT, 1), len(self.layers) + 1):
            getattr(self, "layer%d" % i).train(mode)
        return self

    _stem_weight = _ResNetBody._stem_weight
    forward = _ResNetBody.forward


@registry.BACKBONES.register("resnext")
def resnext():
    return ResNeXt(norm=get_norm())
