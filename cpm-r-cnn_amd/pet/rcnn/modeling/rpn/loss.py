"""RPN loss (counterpart of pet/rcnn/modeling/rpn/loss.py:18-153): IoU(+1) matching with low-quality matches,
visibility / between-threshold discards, 256 samples per image at 50 % positives, BCE-with-logits over the
sample and smooth-L1(beta=1/9, sum) over its positives divided by the sample size."""
import os

import torch
from torch.nn import functional as F

import pet.lib.ops as ops
from pet.lib.ops import smooth_l1_loss
from pet.lib.ops.roi_lists import gt_pack
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
        # CPM_FUSED_GLUE=0 runs the per-image formulation below (the in-tree cross-check of the fused path)
        self.fused_glue = os.environ.get("CPM_FUSED_GLUE", "1") != "0"

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

    def _call_fused(self, anchors, objectness, box_regression, targets):
        """The same loss over the whole batch with no host round trip: one cpm_match_rois launch for every anchor of
        every image (IoU + thresholds + low-quality matches), one batch-wide sampler, and one launch for the regression
        targets, both losses and both gradients over all anchors under the sample masks (identical values and gradients
        to the nonzero()-gathered subsets: a gather's backward scatters into zeros exactly where the mask is zero)."""
        n_img = len(anchors)
        if getattr(anchors, "owner", None) is not None:
            abox, vis, img, per = anchors.owner.batch_pack(anchors)       # cached per (feature, image) sizes
        else:
            abox = torch.cat([a.bbox for per_img in anchors for a in per_img], dim=0)
            vis = torch.cat([a.get_field("visibility") for per_img in anchors for a in per_img], dim=0)
            per = abox.shape[0] // n_img
            img = torch.arange(n_img, device=abox.device).repeat_interleave(per).to(torch.int32)
        assert all(sum(len(a) for a in per_img) == per for per_img in anchors)
        gt_all, _, gt_off, _ = gt_pack(targets)
        m = self.proposal_matcher
        matched, _ = ops.match_rois(abox, img, gt_all, gt_off, m.high_threshold, m.low_threshold,
                                    m.allow_low_quality_matches)
        if self.generate_labels_func is generate_rpn_labels:
            lab = ops.rpn_labels(matched, vis if "not_visibility" in self.discard_cases else None,
                                 "between_thresholds" in self.discard_cases)
        else:
            lab = self.generate_labels_func(matched).to(dtype=torch.float32)
            if "between_thresholds" in self.discard_cases:
                lab = torch.where(matched == Matcher.BETWEEN_THRESHOLDS, -1.0, lab)
            if "not_visibility" in self.discard_cases:
                lab = torch.where(vis, lab, -1.0)
        pos, neg, quota = ops.sample_pos_neg(lab, [per] * n_img, self.fg_bg_sampler.batch_size_per_image,
                                               self.fg_bg_sampler.positive_fraction)
        token = getattr(objectness[0], "_cpm_rpn_sparse", None)
        if token is not None:
            # the head is one autograd node (ops.rpn_head): tell it which anchors this loss sums over -- everywhere else
            # the gradient it receives is exactly zero -- as an index list built on the device
            cap = n_img * self.fg_bg_sampler.batch_size_per_image
            idx, _ = ops.mask_compact(pos, neg, cap)
            ops.set_rpn_sample(token, idx, cap, n_img)
        objectness, box_regression = concat_box_prediction_layers(objectness, box_regression)
        # targets (BoxCoder.encode of the matched gt), smooth-L1 over the positives, BCE over the sample, and the
        # gradients of both: ONE launch (cpm_rpn_loss) instead of ~100 elementwise / reduction kernels
        # (the division by the number of sampled anchors, loss.py:121-126, happens inside the launch too: it reads the
        # sampler's quota on the device -- no sum, no two divisions, no two backward multiplications)
        return ops.rpn_loss(objectness.squeeze(1), box_regression, abox, matched, gt_all, gt_off, pos, neg, per,
                            self.box_coder.weights, cfg.RPN.SMOOTH_L1_BETA, quota=quota.contiguous())

    def __call__(self, anchors, objectness, box_regression, targets):
        if self.fused_glue:
            return self._call_fused(anchors, objectness, box_regression, targets)
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
