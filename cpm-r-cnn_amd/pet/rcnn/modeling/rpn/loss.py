"""RPN loss (counterpart of pet/rcnn/modeling/rpn/loss.py:18-153): IoU(+1) matching with low-quality matches,
visibility / between-threshold discards, 256 samples per image at 50 % positives, BCE-with-logits over the
sample and smooth-L1(beta=1/9, sum) over its positives divided by the sample size."""
import torch
from torch.nn import functional as F

from pet.lib.ops import smooth_l1_loss
from pet.rcnn.core.config import cfg
from pet.rcnn.utils.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from pet.rcnn.utils.matcher import Matcher
from pet.rcnn.utils.misc import concat_box_prediction_layers
from pet.utils.data.structures.boxlist_ops import box_iou_plus1


class RPNLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, box_coder, generate_labels_func):
        self.proposal_matcher = proposal_matcher
        self.fg_bg_sampler = fg_bg_sampler
        self.box_coder = box_coder
        self.generate_labels_func = generate_labels_func
        self.discard_cases = ["not_visibility", "between_thresholds"]

    def prepare_targets(self, anchors, targets):
        """anchors: list (per image) of (bbox [A,4], visibility [A]); targets: list[BoxList]."""
        labels, regression_targets = [], []
        for (abox, vis), t in zip(anchors, targets):
            matched = self.proposal_matcher(box_iou_plus1(t.bbox, abox))
            lab = self.generate_labels_func(matched).to(dtype=torch.float32)
            lab = lab.masked_fill(matched == Matcher.BELOW_LOW_THRESHOLD, 0)
            if "not_visibility" in self.discard_cases:
                lab = lab.masked_fill(~vis, -1)
            if "between_thresholds" in self.discard_cases:
                lab = lab.masked_fill(matched == Matcher.BETWEEN_THRESHOLDS, -1)
            labels.append(lab)
            regression_targets.append(self.box_coder.encode(t.bbox[matched.clamp(min=0)], abox))
        return labels, regression_targets

    def __call__(self, anchors, objectness, box_regression, targets):
        flat = [(torch.cat([a.bbox for a in per_img], 0), torch.cat([a.get_field("visibility") for a in per_img], 0))
                for per_img in anchors]
        labels, regression_targets = self.prepare_targets(flat, targets)
        pos_masks, neg_masks = self.fg_bg_sampler(labels)
        pos = torch.nonzero(torch.cat(pos_masks, dim=0)).squeeze(1)
        neg = torch.nonzero(torch.cat(neg_masks, dim=0)).squeeze(1)
        sampled = torch.cat([pos, neg], dim=0)
        objectness, box_regression = concat_box_prediction_layers(objectness, box_regression)
        objectness = objectness.squeeze()
        labels = torch.cat(labels, dim=0)
        regression_targets = torch.cat(regression_targets, dim=0)
        box_loss = smooth_l1_loss(box_regression[pos], regression_targets[pos], beta=cfg.RPN.SMOOTH_L1_BETA,
                                  reduction="sum") / (sampled.numel())
        objectness_loss = F.binary_cross_entropy_with_logits(objectness[sampled], labels[sampled])
        return objectness_loss, box_loss


def generate_rpn_labels(matched_idxs):
    return matched_idxs >= 0


def make_rpn_loss_evaluator(box_coder):
    R = cfg.RPN
    return RPNLossComputation(Matcher(R.FG_IOU_THRESHOLD, R.BG_IOU_THRESHOLD, allow_low_quality_matches=True),
                              BalancedPositiveNegativeSampler(R.BATCH_SIZE_PER_IMAGE, R.POSITIVE_FRACTION), box_coder,
                              generate_rpn_labels)
