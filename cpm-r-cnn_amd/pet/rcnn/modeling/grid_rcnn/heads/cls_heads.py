"""2-FC classification head, also used as the re-scoring (RSM) head (counterpart of
pet/rcnn/modeling/grid_rcnn/heads/cls_heads.py:13-48): fused-FPN RoIAlign 7x7 -> fc6 -> ReLU -> fc7 -> ReLU."""
import torch.nn as nn

from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry
from pet.rcnn.utils.poolers import Pooler
from pet.utils.net import make_fc


@registry.ROI_CLS_HEADS.register("roi_cls_head")
class roi_cls_head(nn.Module):
    def __init__(self, dim_in, spatial_scale):
        super().__init__()
        G = cfg.GRID_RCNN
        self.dim_in = dim_in[-1]
        res = G.ROI_XFORM_RESOLUTION_CLS
        self.pooler = Pooler(method=G.ROI_XFORM_METHOD, output_size=res, scales=spatial_scale,
                             sampling_ratio=G.ROI_XFORM_SAMPLING_RATIO)
        if G.MLP_HEAD.USE_WS:
            raise ValueError("weight-standardised heads are outside the hot path")
        self.fc6 = make_fc(self.dim_in * res[0] * res[1], G.MLP_HEAD.MLP_DIM, G.MLP_HEAD.USE_BN, G.MLP_HEAD.USE_GN,
                           window=(self.dim_in, res[0], res[1]))
        self.fc7 = make_fc(G.MLP_HEAD.MLP_DIM, G.MLP_HEAD.MLP_DIM, G.MLP_HEAD.USE_BN, G.MLP_HEAD.USE_GN)
        self.dim_out = G.MLP_HEAD.MLP_DIM

    def forward(self, x, proposals):
        x = self.pooler(x, proposals)                 # [R, C, 7, 7], NHWC in memory
        # fc6's result feeds fc7 alone, fc7's result the one Linear of Cls_output (outputs.py): their ReLU gates are
        # applied by those consumers' data-gradient epilogues instead of a pass of their own
        x = self.fc6(x, relu=True, sole_consumer=True)              # flatten folded into the full-window conv
        return self.fc7(x, relu=True, sole_consumer=True)
