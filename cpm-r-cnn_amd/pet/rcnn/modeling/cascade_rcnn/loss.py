"""Sampling and losses of a cascade stage (counterpart of pet/rcnn/modeling/cascade_rcnn/loss.py:15-222):
IoU matcher at the stage's threshold, 512 x 0.25 sampler, cross-entropy + smooth-L1 on the (class-agnostic) deltas,
and on the last stage the ISM IoU regression.  Reference behaviour kept as is: the IoU targets are built from ALL
matched proposals of prepare_targets (before sampling) and paired with the logits through l2_loss's row indexing
(loss.py:55-67,204-206)."""
import torch
from torch.nn import functional as F

from pet.lib.ops import cross_entropy_fused, l2_loss, smooth_l1_loss
from pet.rcnn.core.config import cfg
from pet.rcnn.utils.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from pet.rcnn.utils.box_coder import BoxCoder
from pet.rcnn.utils.matcher import Matcher
from pet.rcnn.utils.misc import cat
from pet.utils.data.structures.boxlist_ops import boxlist_iou


class CascadeRCNNLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, box_coder, cls_agnostic_bbox_reg=False, rescore_on=False, idx=1):
        self.proposal_matcher, self.fg_bg_sampler, self.box_coder = proposal_matcher, fg_bg_sampler, box_coder
        self.cls_agnostic_bbox_reg, self.rescore_on, self.stage = cls_agnostic_bbox_reg, rescore_on, idx

    def prepare_targets(self, proposals, targets):
        labels, regression_targets, self.match_quality_matrixs = [], [], []
        for p, t in zip(proposals, targets):
            q = boxlist_iou(t, p)
            matched = self.proposal_matcher(q)
            gt = t.copy_with_fields("labels")[matched.clamp(min=0)]
            lab = gt.get_field("labels").to(dtype=torch.int64)
            lab = lab.masked_fill(matched == Matcher.BELOW_LOW_THRESHOLD, 0)
            lab = lab.masked_fill(matched == Matcher.BETWEEN_THRESHOLDS, -1)
            labels.append(lab)
            regression_targets.append(self.box_coder.encode(gt.bbox, p.bbox))
            self.match_quality_matrixs.append(q[:, matched >= 0] if cfg.CASCADE_RCNN.IOU_HELPER else None)
        return labels, regression_targets

    def prepare_iou_target(self):
        out = []
        for q in self.match_quality_matrixs:
            assert q is not None
            fg = q.max(dim=0)[0].unsqueeze(1)
            out.append(torch.cat([1 - fg, fg], dim=1))
        return torch.cat(out)

    def subsample(self, proposals, targets):
        labels, regression_targets = self.prepare_targets(proposals, targets)
        pos_masks, neg_masks = self.fg_bg_sampler(labels)
        proposals = list(proposals)
        for lab, reg, p in zip(labels, regression_targets, proposals):
            p.add_field("labels", lab)
            p.add_field("regression_targets", reg)
        for i, (pm, nm) in enumerate(zip(pos_masks, neg_masks)):
            proposals[i] = proposals[i][torch.nonzero(pm | nm).squeeze(1)]
        self._proposals = proposals
        return proposals

    def __call__(self, class_logits, box_regression, iou_logits):
        class_logits, box_regression = cat(class_logits, dim=0), cat(box_regression, dim=0)
        if not hasattr(self, "_proposals"):
            raise RuntimeError("subsample needs to be called before")
        labels = cat([p.get_field("labels") for p in self._proposals], dim=0)
        regression_targets = cat([p.get_field("regression_targets") for p in self._proposals], dim=0)
        classification_loss = cross_entropy_fused(class_logits, labels)
        pos = torch.nonzero(labels > 0).squeeze(1)
        if self.cls_agnostic_bbox_reg:
            map_inds = torch.tensor([4, 5, 6, 7], device=class_logits.device)
        else:
            map_inds = 4 * labels[pos][:, None] + torch.tensor([0, 1, 2, 3], device=class_logits.device)
        box_loss = smooth_l1_loss(box_regression[pos[:, None], map_inds], regression_targets[pos],
                                  beta=cfg.FAST_RCNN.SMOOTH_L1_BETA, reduction="sum") / labels.numel()
        loss_iou = 0
        if cfg.CASCADE_RCNN.IOU_HELPER and self.stage == cfg.CASCADE_RCNN.NUM_STAGE - 1:
            loss_iou = l2_loss(iou_logits, self.prepare_iou_target())
        return classification_loss, box_loss, loss_iou


def box_loss_evaluator(idx):
    C = cfg.CASCADE_RCNN
    matcher = Matcher(C.FG_IOU_THRESHOLD[idx], C.BG_IOU_THRESHOLD[idx], allow_low_quality_matches=False)
    sampler = BalancedPositiveNegativeSampler(cfg.FAST_RCNN.BATCH_SIZE_PER_IMAGE, cfg.FAST_RCNN.POSITIVE_FRACTION)
    return CascadeRCNNLossComputation(matcher, sampler, BoxCoder(weights=C.BBOX_REG_WEIGHTS[idx]),
                                      cfg.MODEL.CLS_AGNOSTIC_BBOX_REG, C.RESCORE_ON, idx)
