"""Offset-regression Cascade R-CNN box head, with the ISM (IoU branch on the last stage) and RSM (re-scoring head)
variants of cfgs/rcnn/mscoco/cascade/{ISM,RSM,ISM+RSM} (counterpart of
pet/rcnn/modeling/cascade_rcnn/cascade_rcnn.py:16-143).  Same kernels as the CPM head: fused-FPN
RoIAlign, fc6 as a full-window MFMA conv, the FC stack on the implicit-GEMM kernel."""
import copy

import torch
from torch import nn

from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry
from pet.rcnn.modeling.cascade_rcnn import heads, outputs  # noqa: F401  (register)
from pet.rcnn.modeling.cascade_rcnn.inference import box_post_processor
from pet.rcnn.modeling.cascade_rcnn.loss import box_loss_evaluator


class CascadeRCNN(nn.Module):
    def __init__(self, dim_in, spatial_scale):
        super().__init__()
        C = cfg.CASCADE_RCNN
        self.num_stage, self.test_stage = C.NUM_STAGE, C.TEST_STAGE
        self.stage_loss_weights, self.test_ensemble = C.STAGE_WEIGHTS, C.TEST_ENSEMBLE
        head = registry.ROI_CASCADE_HEADS[C.ROI_BOX_HEAD]
        output = registry.ROI_CASCADE_OUTPUTS[C.ROI_BOX_OUTPUT]
        for stage in range(1, self.num_stage + 1):
            setattr(self, "Box_Head_%d" % stage, head(dim_in, spatial_scale))
            setattr(self, "Output_%d" % stage, output(getattr(self, "Box_Head_%d" % stage).dim_out, stage - 1))
        if C.RESCORE_ON:
            import pet.rcnn.modeling.grid_rcnn.heads  # noqa: F401  (registers roi_cls_head)
            import pet.rcnn.modeling.grid_rcnn.outputs  # noqa: F401
            from pet.rcnn.modeling.grid_cascade_rcnn.inference import post_processor
            from pet.rcnn.modeling.grid_cascade_rcnn.loss import loss_evaluator
            self.Head_rescore = registry.ROI_CLS_HEADS[cfg.GRID_RCNN.ROI_CLS_HEAD](dim_in, spatial_scale)
            self.Output_rescore = registry.ROI_CLS_OUTPUTS[cfg.GRID_RCNN.ROI_CLS_OUTPUT](self.Head_rescore.dim_out)
            self.rescore_loss_evaluator = loss_evaluator(type="cls")
            self.cls_post_processor = post_processor(type="cls")
            self.cls_init_proposals = None

    def forward(self, conv_features, proposals, targets=None):
        C = cfg.CASCADE_RCNN
        all_loss, ms_scores, per_loss_iou, x = {}, [], 0, None
        for i in range(self.num_stage):
            head, output = getattr(self, "Box_Head_%d" % (i + 1)), getattr(self, "Output_%d" % (i + 1))
            evaluator = box_loss_evaluator(i)
            if self.training:
                with torch.no_grad():
                    proposals = evaluator.subsample(proposals, targets)
                    if i == 0:
                        self.cls_init_proposals = copy.deepcopy(proposals)
            x = head(conv_features, proposals)
            class_logits, box_regression, iou_logits = output(x)
            ms_scores.append(class_logits)
            if not self.training:
                post = box_post_processor(i, is_train=False)
                if i < self.test_stage - 1:
                    proposals = post((class_logits, box_regression), proposals, iou_logits=iou_logits)
                    continue
                if self.test_ensemble:
                    assert len(ms_scores) == self.test_stage
                    class_logits = sum(ms_scores) / self.test_stage
                return x, post((class_logits, box_regression), proposals, iou_logits=iou_logits), {}
            loss_cls, loss_box, per_loss_iou = evaluator([class_logits], [box_regression], iou_logits)
            all_loss["s%d_cls_loss" % (i + 1)] = loss_cls * self.stage_loss_weights[i]
            all_loss["s%d_bbox_loss" % (i + 1)] = loss_box * self.stage_loss_weights[i]
            if i < self.num_stage - 1:
                with torch.no_grad():
                    proposals = box_post_processor(i, is_train=True)((class_logits, box_regression), proposals, targets,
                                                                     iou_logits=iou_logits)
        if C.IOU_HELPER:
            all_loss["loss_iou_%d" % self.num_stage] = per_loss_iou * C.IOU_LOSS_WEIGHT
        if C.RESCORE_ON and self.training:
            proposals, loss_rescore = self._forward_train_rescore(conv_features, self.cls_init_proposals, proposals,
                                                                  targets)
            all_loss.update(loss_rescore)
        return x, proposals, all_loss

    def _forward_train_rescore(self, features, cls_proposals, last_proposals, targets):
        from pet.rcnn.modeling.grid_cascade_rcnn.grid_cascade_rcnn import get_full_sample_boxes
        assert cls_proposals is not None
        with torch.no_grad():
            proposals = get_full_sample_boxes(cls_proposals, last_proposals)
            proposals = self.rescore_loss_evaluator.subsample(proposals, targets)
        logits = self.Output_rescore(self.Head_rescore(features, proposals))
        loss = self.rescore_loss_evaluator([logits]) * cfg.CASCADE_RCNN.RESCORE_LOSS_WEIGHT
        return proposals, dict(loss_rescore=loss)
