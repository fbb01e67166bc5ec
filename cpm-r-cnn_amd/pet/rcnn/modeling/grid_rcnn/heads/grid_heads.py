"""CMM per-point feature stack (counterpart of pet/rcnn/modeling/grid_rcnn/heads/grid_heads.py:14-160, the
FUSED_ON=False path every BASELINE config uses): fused-FPN RoIAlign 14x14, then 8 x [conv3x3 (first one
stride 2) -> GroupNorm(4*points) -> ReLU] at 64*points channels.  58 % of the model's forward MACs."""
import numpy as np
from torch import nn

import pet.lib.ops as ops
from pet.lib.ops import conv as conv_ops
from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry
from pet.rcnn.utils.poolers import Pooler


@registry.ROI_GRID_HEADS.register("roi_grid_head")
class roi_grid_head(nn.Module):
    def __init__(self, dim_in, spatial_scale, stage):
        super().__init__()
        G = cfg.GRID_RCNN
        if G.FUSED_ON or G.OFFSET_ON:
            raise ValueError("GRID_RCNN.FUSED_ON / OFFSET_ON variants are outside the hot path (BASELINE cfgs set False)")
        self.grid_points = G.CASCADE_MAPPING_OPTION.GRID_NUM[stage] if G.CASCADE_MAPPING_ON else G.GRID_POINTS
        self.roi_feat_size = G.ROI_FEAT_SIZE
        self.num_convs = G.GRID_HEAD.NUM_CONVS
        self.point_feat_channels = G.GRID_HEAD.POINT_FEAT_CHANNELS
        self.conv_out_channels = self.point_feat_channels * self.grid_points
        self.dim_in = dim_in[-1]
        assert self.grid_points >= 4
        self.grid_size = int(np.sqrt(self.grid_points))
        if self.grid_size * self.grid_size != self.grid_points:
            raise ValueError("grid_points must be a square number")
        if not isinstance(self.roi_feat_size, int):
            raise ValueError("Only square RoIs are supporeted in Grid R-CNN")
        self.whole_map_size = self.roi_feat_size * 4
        blocks = []
        for i in range(self.num_convs):
            blocks.append(nn.Sequential(
                ops.Conv2d(self.dim_in if i == 0 else self.conv_out_channels, self.conv_out_channels, kernel_size=3,
                           stride=2 if i == 0 else 1, padding=1),
                ops.GroupNorm(4 * self.grid_points, self.conv_out_channels, eps=1e-5),
                ops.ReLU(inplace=True)))
        self.convs = nn.Sequential(*blocks)
        scales = [spatial_scale[0]] if G.FINEST_LEVEL_ROI else spatial_scale
        self.pooler = Pooler(method=G.ROI_XFORM_METHOD, output_size=G.ROI_XFORM_RESOLUTION_GRID, scales=scales,
                             sampling_ratio=G.ROI_XFORM_SAMPLING_RATIO)
        self.dim_out = dim_in

    def forward(self, features, proposals):
        x = self.pooler(features, proposals)
        assert x.shape[-1] == x.shape[-2] == self.roi_feat_size
        # conv -> GroupNorm+ReLU x 8: one autograd node / one native call per direction when training on the flat
        # gradient buffer (pet/lib/ops/conv.py: conv_gn_stack), layer by layer otherwise
        x = conv_ops.conv_gn_stack(x, [blk[0] for blk in self.convs], [blk[1] for blk in self.convs])
        return x, None
