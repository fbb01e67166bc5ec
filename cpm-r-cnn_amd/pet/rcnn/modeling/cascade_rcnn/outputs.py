"""Class scores, box deltas and (last stage, ISM) the IoU branch of a cascade stage (counterpart of
pet/rcnn/modeling/cascade_rcnn/outputs.py:13-57)."""
import torch.nn as nn
import torch.nn.init as init

import pet.lib.ops as ops
from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry


@registry.ROI_CASCADE_OUTPUTS.register("Box_output")
class Box_output(nn.Module):
    def __init__(self, dim_in, stage):
        super().__init__()
        self.stage, self.dim_in = stage, dim_in
        self.cls_score = ops.Linear(dim_in, cfg.MODEL.NUM_CLASSES)
        self.bbox_pred = ops.Linear(dim_in, 4 * (2 if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG else cfg.MODEL.NUM_CLASSES))
        self.has_iou = bool(cfg.CASCADE_RCNN.IOU_HELPER and stage == cfg.CASCADE_RCNN.NUM_STAGE - 1)
        if self.has_iou:
            self.iou_fc1 = ops.Linear(dim_in, 1024)
            self.iou_fc2 = ops.Linear(1024, 1024)
            self.iou_pred = ops.Linear(1024, 2)
            init.normal_(self.iou_pred.weight, std=0.01)
            init.constant_(self.iou_pred.bias, 0)
        init.normal_(self.cls_score.weight, std=0.01)
        init.constant_(self.cls_score.bias, 0)
        init.normal_(self.bbox_pred.weight, std=0.001)
        init.constant_(self.bbox_pred.bias, 0)

    def forward(self, x):
        if x.ndimension() == 4:
            x = x.mean(dim=(2, 3))
        cls_score = self.cls_score(x)
        bbox_pred = self.bbox_pred(x)
        iou_logits = None
        if self.has_iou:
            t = self.iou_fc1(x, relu=True)
            t = self.iou_fc2(t, relu=True)
            iou_logits = self.iou_pred(t)
        return cls_score, bbox_pred, iou_logits
