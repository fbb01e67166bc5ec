"""2-FC box head of the offset-regression cascade (counterpart of
pet/rcnn/modeling/cascade_rcnn/heads/mlp_heads.py:12-48): fused-FPN RoIAlign -> fc6 -> ReLU -> fc7 -> ReLU."""
import torch.nn as nn

from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry
from pet.rcnn.utils.poolers import Pooler
from pet.utils.net import make_fc


@registry.ROI_CASCADE_HEADS.register("roi_2mlp_head")
class roi_2mlp_head(nn.Module):
    def __init__(self, dim_in, spatial_scale):
        super().__init__()
        F_ = cfg.FAST_RCNN
        self.dim_in = dim_in[-1]
        res = F_.ROI_XFORM_RESOLUTION
        self.pooler = Pooler(method=F_.ROI_XFORM_METHOD, output_size=res, scales=spatial_scale,
                             sampling_ratio=F_.ROI_XFORM_SAMPLING_RATIO)
        if F_.MLP_HEAD.USE_WS:
            raise ValueError("weight-standardised heads are outside the built path")
        self.fc6 = make_fc(self.dim_in * res[0] * res[1], F_.MLP_HEAD.MLP_DIM, F_.MLP_HEAD.USE_BN, F_.MLP_HEAD.USE_GN,
                           window=(self.dim_in, res[0], res[1]))
        self.fc7 = make_fc(F_.MLP_HEAD.MLP_DIM, F_.MLP_HEAD.MLP_DIM, F_.MLP_HEAD.USE_BN, F_.MLP_HEAD.USE_GN)
        self.dim_out = F_.MLP_HEAD.MLP_DIM

    def forward(self, x, proposals):
        x = self.pooler(x, proposals)                 # [R, C, h, w], NHWC in memory
        x = self.fc6(x, relu=True)                    # flatten folded into the full-window conv
        return self.fc7(x, relu=True)
