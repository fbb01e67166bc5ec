"""Feature Pyramid Network (counterpart of pet/rcnn/modeling/fpn/FPN.py:15-139, body "fpn").

Each lateral 1x1 conv adds the nearest-2x upsampled coarser map in its epilogue (res_mode=1), so the
reference's interpolate + add kernels disappear; P6 is the stride-2 subsample of P5 (MaxPool2d(1, 2))."""
import torch.nn as nn

import pet.lib.ops as ops

from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry
from pet.utils.net import make_conv


@registry.FPN_BODY.register("fpn")
class fpn(nn.Module):
    def __init__(self, dim_in, spatial_scale):
        super().__init__()
        F = cfg.FPN
        if F.USE_LITE or F.USE_BN or F.USE_GN or F.USE_WS or F.EXTRA_CONV_LEVELS:
            raise ValueError("only the plain FPN (no lite/BN/GN/WS/extra convs) is on the hot path")
        self.dim_in = dim_in[-1]
        self.spatial_scale = list(spatial_scale)
        fpn_dim = F.DIM
        min_level, max_level = get_min_max_levels()
        self.num_backbone_stages = len(dim_in) - (min_level - F.LOWEST_BACKBONE_LVL)
        self.p5_in = make_conv(self.dim_in, fpn_dim, kernel=1)
        self.p5_out = make_conv(fpn_dim, fpn_dim, kernel=3)
        self.fpn_in = nn.ModuleList()
        self.fpn_out = nn.ModuleList()
        for i in range(self.num_backbone_stages - 1):
            self.fpn_in.append(make_conv(dim_in[-i - 2], fpn_dim, kernel=1))
            self.fpn_out.append(make_conv(fpn_dim, fpn_dim, kernel=3))
        self.dim_in = fpn_dim
        self.has_p6 = max_level == F.HIGHEST_BACKBONE_LVL + 1
        if self.has_p6:
            self.maxpool_p6 = nn.MaxPool2d(kernel_size=1, stride=2, padding=0)
            self.spatial_scale.append(self.spatial_scale[-1] * 0.5)
        num_roi_levels = F.ROI_MAX_LEVEL - F.ROI_MIN_LEVEL + 1
        self.spatial_scale = self.spatial_scale[:num_roi_levels]
        self.dim_out = [self.dim_in for _ in range(num_roi_levels)]
        self._init_weights()

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        px = ops.mark_shared_grad(self.p5_in(x[-1]))
        # a level's 3x3 output conv reads its own px only: it runs on the second stream beside the lateral chain of the
        # finer levels (ops.fwd_fork); the finest level's -- the largest, and the last -- stays on the compute stream
        n_lat = self.num_backbone_stages - 1
        sided = False

        def out_conv(conv, t, last):
            nonlocal sided
            if not last and ops.fwd_fork(t, 2):
                sided = True
                with ops.fwd_side(t):
                    return conv(t)
            return conv(t)
        outs = [out_conv(self.p5_out, px, n_lat == 0)]
        for i in range(self.num_backbone_stages - 1):
            c = x[-i - 2]
            if tuple(c.shape[2:]) != tuple(px.shape[2:]):
                assert c.shape[2] == 2 * px.shape[2] and c.shape[3] == 2 * px.shape[3], \
                    "FPN levels must be exactly 2x apart (input size divisible by 32)"
                px = self.fpn_in[i](c, residual=px, res_mode=1)      # lateral + nearest-2x(top), fused
            else:
                px = self.fpn_in[i](c, residual=px, res_mode=0)
            ops.mark_shared_grad(px)   # consumers: this level's output conv and the next lateral's top-down residual
            outs.insert(0, out_conv(self.fpn_out[i], px, i == n_lat - 1))
        if sided:
            ops.fwd_join(px)
        for o in outs:                 # consumers: RPN head conv, the RoIAlign calls (and P6's subsample, created first)
            ops.mark_shared_grad(o)
        if self.has_p6:
            outs.append(outs[-1][:, :, ::2, ::2].contiguous(memory_format=__import__("torch").channels_last))
        return outs


def get_min_max_levels():
    F = cfg.FPN
    min_level, max_level = F.LOWEST_BACKBONE_LVL, F.HIGHEST_BACKBONE_LVL
    if F.MULTILEVEL_RPN and not F.MULTILEVEL_ROIS:
        max_level, min_level = F.RPN_MAX_LEVEL, F.RPN_MIN_LEVEL
    if not F.MULTILEVEL_RPN and F.MULTILEVEL_ROIS:
        max_level, min_level = F.ROI_MAX_LEVEL, F.ROI_MIN_LEVEL
    if F.MULTILEVEL_RPN and F.MULTILEVEL_ROIS:
        max_level = max(F.RPN_MAX_LEVEL, F.ROI_MAX_LEVEL)
        min_level = min(F.RPN_MIN_LEVEL, F.ROI_MIN_LEVEL)
    return min_level, max_level
