"""Names of the reference's operator package that no CPM R-CNN configuration reaches
(/root/reference/pet/lib/ops/__init__.py:1-30; SURVEY 8b: every name must import because callers type-check against
them, e.g. `isinstance(m, ops.MixtureBatchNorm2d)` at pet/models/imagenet/resnet.py:285).

They exist as importable types / callables and fail loudly on USE: the hot path never constructs them, and
a silent torch fallback would hide a missing kernel.  FrozenBatchNorm2d and Scale are plain parameter holders with
one-line arithmetic and are real."""
import torch
from torch import nn


def _off_path_module(name, why="is outside the CPM R-CNN hot path"):
    def __init__(self, *args, **kwargs):
        raise RuntimeError("pet.lib.ops.%s %s and is not provided by cpm-r-cnn_amd" % (name, why))
    return type(name, (nn.Module,), {"__init__": __init__, "__doc__": "off-path placeholder for ops.%s" % name})


def _off_path_function(name):
    def fn(*args, **kwargs):
        raise RuntimeError("pet.lib.ops.%s is outside the CPM R-CNN hot path and is not provided by cpm-r-cnn_amd"
                           % name)
    fn.__name__ = name
    return fn


_MODULES = ("IOULoss", "BoundedIoULoss", "MaskIOULoss", "DICELoss", "SigmoidFocalLoss", "LovaszHinge", "LovaszSoftmax",
            "LabelSmoothing", "NaiveSyncBatchNorm", "Conv2dSamePadding", "Conv2dWS", "SplAtConv2d",
            "ModulatedDeformConv", "ModulatedDeformConvPack", "L2Norm", "MixtureBatchNorm2d", "MixtureGroupNorm", "Mish",
            "H_Swish", "H_Sigmoid", "Swish", "SwishX", "DropBlock2D", "SeConv2d", "GlobalContextBlock", "ECA",
            "ROIAlignRotated", "ROIPool")
_FUNCTIONS = ("nms_rotated", "poly_nms", "box_iou_rotated", "smooth_l1_loss_LW", "equalization_loss",
              "lovasz_softmax_loss", "roi_align_rotated", "roi_pool")

for _n in _MODULES:
    globals()[_n] = _off_path_module(_n)
for _n in _FUNCTIONS:
    globals()[_n] = _off_path_function(_n)


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters (buffers `weight`, `bias`, `running_mean`,
    `running_var`, the reference's names): y = x * scale + shift.  The CPM configs fold every norm into
    AffineChannel2d instead (pet/utils/net.py), so this never runs on the hot path."""

    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


class Scale(nn.Module):
    """A learnable scalar factor."""

    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([float(init_value)]))

    def forward(self, x):
        return x * self.scale


__all__ = list(_MODULES) + list(_FUNCTIONS) + ["FrozenBatchNorm2d", "Scale"]
