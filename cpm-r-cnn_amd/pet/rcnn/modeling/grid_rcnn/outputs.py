"""Heat-map output + ISM, and the class-score output (counterpart of pet/rcnn/modeling/grid_rcnn/outputs.py:13-104).

Grid_output: grouped ConvTranspose2d(k4,s2,p1) -> GroupNorm(points)+ReLU -> grouped ConvTranspose2d -> [R,P,28,28]
logits; on the last stage the ISM branch iou_fc1 -> ReLU -> iou_fc2 -> ReLU -> iou_pred on the same features.
The transposed convs run on the implicit-GEMM kernel in data-gradient mode (bias fused)."""
import torch.nn.init as init
from torch import nn

import pet.lib.ops as ops
from pet.lib.ops import conv as conv_ops
from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry
from pet.utils.net import make_fc


@registry.ROI_GRID_OUTPUTS.register("Grid_output")
class Grid_output(nn.Module):
    def __init__(self, dim_in, stage):
        super().__init__()
        G = cfg.GRID_RCNN
        if G.FUSED_ON or G.OFFSET_ON or G.SE_ON:
            raise ValueError("GRID_RCNN.FUSED_ON / OFFSET_ON / SE_ON variants are outside the hot path")
        self.stage = stage
        self.dim_in = dim_in[-1]
        self.grid_points = G.CASCADE_MAPPING_OPTION.GRID_NUM[stage] if G.CASCADE_MAPPING_ON else G.GRID_POINTS
        self.point_feat_channels = G.GRID_HEAD.POINT_FEAT_CHANNELS
        self.conv_out_channels = self.point_feat_channels * self.grid_points
        self.norm1 = ops.GroupNorm(self.grid_points, self.conv_out_channels)
        self.deconv_1 = ops.ConvTranspose2d(self.conv_out_channels, self.conv_out_channels, kernel_size=4, stride=2,
                                            padding=1, groups=self.grid_points)
        self.deconv_2 = ops.ConvTranspose2d(self.conv_out_channels, self.grid_points, kernel_size=4, stride=2,
                                            padding=1, groups=self.grid_points)
        self.has_iou = bool(G.IOU_HELPER and stage == G.CASCADE_MAPPING_OPTION.STAGE_NUM - 1)
        if self.has_iou:
            res = G.ROI_XFORM_RESOLUTION_CLS
            self.iou_fc1 = make_fc(self.conv_out_channels * res[0] * res[1], 1024,
                                   window=(self.conv_out_channels, res[0], res[1]))
            self.iou_fc2 = make_fc(1024, 1024)
            self.iou_pred = ops.Linear(1024, 2)
            init.normal_(self.iou_pred.weight, std=0.01)
            init.constant_(self.iou_pred.bias, 0)

    def forward(self, x, x_so=None):
        x1 = self.deconv_1(x)
        x1 = self.norm1(x1, relu=True)
        heat = self.deconv_2(x1)
        iou_logits = None
        if self.has_iou:
            # iou_fc1 -> ReLU -> iou_fc2 -> ReLU -> iou_pred: one native call per direction in training
            iou_logits = conv_ops.mlp_chain(x, [self.iou_fc1, self.iou_fc2, self.iou_pred], self)
        return dict(fused=None, unfused=heat), iou_logits


@registry.ROI_CLS_OUTPUTS.register("Cls_output")
class Cls_output(nn.Module):
    def __init__(self, dim_in):
        super().__init__()
        self.dim_in = dim_in
        self.cls_score = ops.Linear(self.dim_in, cfg.MODEL.NUM_CLASSES)
        init.normal_(self.cls_score.weight, std=0.01)
        init.constant_(self.cls_score.bias, 0)

    def forward(self, x):
        if x.ndimension() == 4:
            x = x.mean(dim=(2, 3))
        return self.cls_score(x)
