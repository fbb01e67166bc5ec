"""Per-stage post-processing of the offset-regression cascade (counterpart of
pet/rcnn/modeling/cascade_rcnn/inference.py:14-201).  Intermediate stages decode the class-agnostic deltas into the
next stage's proposals (training: drop degenerate boxes and previous-stage gts, then append the gts); the final stage
repeats the one box per class with softmax scores, optionally multiplied by the ISM IoU score."""
import torch
import torch.nn.functional as F
from torch import nn

from pet.rcnn.core.config import cfg
from pet.rcnn.utils.box_coder import BoxCoder
from pet.utils.data.structures.bounding_box import BoxList
from pet.utils.data.structures.boxlist_ops import cat_boxlist


class PostProcessor(nn.Module):
    def __init__(self, score_thresh=0.05, nms=0.5, detections_per_img=100, box_coder=None,
                 cls_agnostic_bbox_reg=False, is_repeat=False):
        super().__init__()
        self.score_thresh, self.nms, self.detections_per_img = score_thresh, nms, detections_per_img
        self.box_coder = box_coder if box_coder is not None else BoxCoder(weights=(10., 10., 5., 5.))
        self.cls_agnostic_bbox_reg, self.is_repeat = cls_agnostic_bbox_reg, is_repeat

    def forward(self, x, boxes, targets=None, iou_logits=None):
        class_logits, box_regression = x
        class_prob = F.softmax(class_logits, -1)
        image_shapes = [b.size for b in boxes]
        per_image = [len(b) for b in boxes]
        concat = torch.cat([b.bbox for b in boxes], dim=0)
        if self.cls_agnostic_bbox_reg:
            box_regression = box_regression[:, -4:]
        proposals = self.box_coder.decode(box_regression.reshape(sum(per_image), -1), concat)
        iou_score = None
        if self.cls_agnostic_bbox_reg:
            if self.is_repeat:
                if cfg.CASCADE_RCNN.IOU_HELPER:
                    assert iou_logits is not None
                    iou_score = iou_logits[:, 1]
                proposals = proposals.repeat(1, class_prob.shape[1])
            else:
                return self.refine(boxes, targets, proposals.split(per_image, dim=0))
        out = []
        for prob, bx, shape in zip(class_prob.split(per_image, dim=0), proposals.split(per_image, dim=0), image_shapes):
            out.append(self.prepare_boxlist(bx, prob, shape, iou_score).clip_to_image(remove_empty=False))
        return out

    def refine(self, boxes, targets, proposals):
        out = []
        if targets is not None:
            for box, t, p in zip(boxes, targets, proposals):
                keep = self._filter_boxes(p, box, t)
                for field in list(box.fields()):
                    box.add_field(field, box.get_field(field)[keep])
                box.bbox = p[keep]
                out.append(box)
            return self.add_gt_proposals(out, targets)
        for box, p in zip(boxes, proposals):
            box.bbox = p
            out.append(box)
        return out

    @staticmethod
    def _filter_boxes(bbox, last, gt):
        """positive width and height, and the previous box is not (component-wise) a gt box (inference.py:121-133)."""
        last_bbox = last.bbox
        ws = bbox[:, 2] - bbox[:, 0] + 1
        hs = bbox[:, 3] - bbox[:, 1] + 1
        for i in range(gt.bbox.shape[0]):
            last_bbox = torch.where(last_bbox == gt.bbox[i], torch.full_like(last_bbox, -1), last_bbox)
        s = last_bbox[:, 0] + last_bbox[:, 1] + last_bbox[:, 2] + last_bbox[:, 3]
        return torch.nonzero((ws > 0) & (hs > 0) & (s > 0)).squeeze(1)

    @staticmethod
    def add_gt_proposals(proposals, targets):
        device = proposals[0].bbox.device
        out = []
        for p, t in zip(proposals, targets):
            gt = t.copy_with_fields(["labels"])
            gt.add_field("objectness", torch.ones(len(gt), device=device))
            gt.add_field("regression_targets", torch.zeros((len(gt), 4), device=device))
            out.append(cat_boxlist((p, gt)))
        return out

    @staticmethod
    def prepare_boxlist(boxes, scores, image_shape, iou_score=None):
        boxes = boxes.reshape(-1, 4)
        scores = scores.reshape(-1)
        if iou_score is not None:
            # inference.py:176-177: [N] -> repeat(1, 81) -> [1, 81 N] -> [81, N] -> T -> row i = 81 copies of iou_i
            iou_score = iou_score.repeat(1, 81).reshape(81, -1).T.reshape(-1)
            if cfg.CASCADE_RCNN.IOU_HELPER_MERGE:
                scores = scores * iou_score
        bl = BoxList(boxes, image_shape, mode="xyxy")
        bl.add_field("scores", scores)
        return bl


def box_post_processor(idx, is_train=True):
    C = cfg.CASCADE_RCNN
    final_test, final_train = idx == C.TEST_STAGE - 1, idx == C.NUM_STAGE - 1
    return PostProcessor(cfg.FAST_RCNN.SCORE_THRESH, cfg.FAST_RCNN.NMS, cfg.FAST_RCNN.DETECTIONS_PER_IMG,
                         BoxCoder(weights=C.BBOX_REG_WEIGHTS[idx]), cfg.MODEL.CLS_AGNOSTIC_BBOX_REG,
                         (is_train and final_train) or (not is_train and final_test))
