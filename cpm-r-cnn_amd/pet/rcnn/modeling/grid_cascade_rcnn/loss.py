"""Losses of the CPM head (counterpart of pet/rcnn/modeling/grid_cascade_rcnn/loss.py:18-369).

CLSLossComputation is the usual Fast R-CNN sampler + cross-entropy.  GridLossComputation builds the per-point
heat-map targets.  The reference rasterises them in a Python triple loop on the CPU (loss.py:213-249, R*9*9
iterations + .cpu()/.cuda() round trips per stage); here the same arithmetic (fp32, same operation order, int()
truncation toward zero, radius-1 disc, 28x28 sub-region crop) is evaluated as whole-tensor ops on the device."""
import os

import numpy as np
import torch
from torch.nn import functional as F

import pet.lib.ops as ops
from pet.lib.ops import l2_loss
from pet.rcnn.core.config import cfg
from pet.rcnn.modeling.grid_rcnn.loss import calc_sub_regions
from pet.rcnn.utils.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from pet.rcnn.utils.matcher import Matcher
from pet.rcnn.utils.misc import cat
from pet.utils.data.structures.boxlist_ops import boxlist_iou


class CLSLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, cls_agnostic_bbox_reg=False):
        self.proposal_matcher = proposal_matcher
        self.fg_bg_sampler = fg_bg_sampler
        self.cls_agnostic_bbox_reg = cls_agnostic_bbox_reg
        # CPM_FUSED_GLUE=0 runs the per-image formulation (the in-tree cross-check of the fused path)
        self.fused_glue = os.environ.get("CPM_FUSED_GLUE", "1") != "0"
        self._packed_labels = None

    def set_packed_sample(self, boxlists, labels):
        """the sample as produced on the device (pet/lib/ops/roi_lists.py): per-image views + the packed labels"""
        self._proposals, self._packed_labels = boxlists, labels

    def prepare_targets(self, proposals, targets):
        labels = []
        for p, t in zip(proposals, targets):
            matched = self.proposal_matcher(boxlist_iou(t, p))
            lab = t.get_field("labels")[matched.clamp(min=0)].to(dtype=torch.int64)
            lab = lab.masked_fill(matched == Matcher.BELOW_LOW_THRESHOLD, 0)
            lab = lab.masked_fill(matched == Matcher.BETWEEN_THRESHOLDS, -1)
            labels.append(lab)
        return labels

    def _subsample_fused(self, proposals, targets):
        """prepare_targets + sampler + selection for all images at once: one cpm_match_rois launch, one batch-wide
        sampler, ONE host round trip (the selection mask) instead of one per image and BoxList field."""
        from pet.utils.data.structures.bounding_box import BoxList
        n_img = len(proposals)
        counts = [len(p) for p in proposals]
        dev = proposals[0].bbox.device
        rois = torch.cat([p.bbox for p in proposals], dim=0)
        img_h = np.repeat(np.arange(n_img), counts)
        img = torch.from_numpy(img_h.astype(np.int32)).pin_memory().to(dev, non_blocking=True)
        gt_off_h = np.concatenate([[0], np.cumsum([len(t) for t in targets])])
        gt_off = torch.from_numpy(gt_off_h.astype(np.int32)).pin_memory().to(dev, non_blocking=True)
        base = torch.from_numpy(gt_off_h[img_h].astype(np.int64)).pin_memory().to(dev, non_blocking=True)
        gt_all = torch.cat([t.bbox for t in targets], dim=0)
        gt_labels = torch.cat([t.get_field("labels") for t in targets], dim=0).to(torch.int64)
        m = self.proposal_matcher
        matched, _ = ops.match_rois(rois, img, gt_all, gt_off, m.high_threshold, m.low_threshold,
                                    m.allow_low_quality_matches)
        lab = gt_labels[matched.clamp(min=0) + base]
        lab = torch.where(matched == Matcher.BELOW_LOW_THRESHOLD, 0, lab)
        lab = torch.where(matched == Matcher.BETWEEN_THRESHOLDS, -1, lab)
        pos, neg, _ = ops.sample_pos_neg(lab, counts, self.fg_bg_sampler.batch_size_per_image,
                                           self.fg_bg_sampler.positive_fraction)
        # the one host round trip: the selection mask, carrying the labels of the selected rows along (-2 = not
        # selected) so that later label-driven selections (positives for the grid branch, negatives for the RSM
        # sample) are made on the host without another device query
        lab_h = torch.where(pos | neg, lab, -2).cpu().numpy()
        idx_h = np.flatnonzero(lab_h != -2)
        new_counts = np.bincount(img_h[idx_h], minlength=n_img).tolist()
        idx = torch.from_numpy(idx_h).pin_memory().to(dev, non_blocking=True)
        fields = {f: torch.cat([p.get_field(f) for p in proposals], dim=0)[idx] for f in proposals[0].fields()
                  if f != "labels"}
        fields["labels"] = lab[idx]
        out, o = [], 0
        sel = rois[idx]
        for i in range(n_img):
            bl = BoxList(sel[o:o + new_counts[i]], proposals[i].size, proposals[i].mode)
            for f, v in fields.items():
                bl.add_field(f, v[o:o + new_counts[i]])
            bl.host_labels = lab_h[idx_h[o:o + new_counts[i]]]       # numpy copy of the "labels" field
            out.append(bl)
            o += new_counts[i]
        self._proposals = out
        return out

    def subsample(self, proposals, targets):
        self._packed_labels = None
        if self.fused_glue and proposals[0].bbox.is_cuda:
            return self._subsample_fused(proposals, targets)
        labels = self.prepare_targets(proposals, targets)
        pos_masks, neg_masks = self.fg_bg_sampler(labels)
        proposals = list(proposals)
        for lab, p in zip(labels, proposals):
            p.add_field("labels", lab)
        for i, (pm, nm) in enumerate(zip(pos_masks, neg_masks)):
            proposals[i] = proposals[i][torch.nonzero(pm | nm).squeeze(1)]
        self._proposals = proposals
        return proposals

    def __call__(self, class_logits):
        class_logits = cat(class_logits, dim=0)
        if not hasattr(self, "_proposals"):
            raise RuntimeError("subsample needs to be called before")
        labels = self._packed_labels
        if labels is None:
            labels = cat([p.get_field("labels") for p in self._proposals], dim=0)
        # value and gradient from one launch on the GPU (cpm_softmax_ce); F.cross_entropy elsewhere
        return ops.cross_entropy_fused(class_logits, labels)


def grid_targets(pos_bboxes, pos_gt_bboxes, mapping_ratio, grid_points, map_size, radius, sub_regions):
    """[R,4] RoIs and their matched gts -> [R, P, half, half] 0/1 targets (loss.py:178-258)."""
    gs = int(np.sqrt(grid_points))
    half = map_size // 4 * 2
    dev = pos_bboxes.device
    b, g = pos_bboxes, pos_gt_bboxes
    x1 = b[:, 0] - mapping_ratio * ((b[:, 2] - b[:, 0]) / 2)
    y1 = b[:, 1] - mapping_ratio * ((b[:, 3] - b[:, 1]) / 2)
    x2 = b[:, 2] + mapping_ratio * ((b[:, 2] - b[:, 0]) / 2)
    y2 = b[:, 3] + mapping_ratio * ((b[:, 3] - b[:, 1]) / 2)
    bw, bh = (x2 - x1).unsqueeze(-1), (y2 - y1).unsqueeze(-1)
    ok = ~((bw <= gs) | (bh <= gs))                                            # "ignore small bboxes"
    j = torch.arange(grid_points, device=dev)
    fx = (1 - torch.div(j, gs, rounding_mode="floor").double() / (gs - 1)).float()[None]
    fy = (1 - (j % gs).double() / (gs - 1)).float()[None]
    gx = fx * g[:, 0:1] + (1 - fx) * g[:, 2:3]                                 # [R, P]
    gy = fy * g[:, 1:2] + (1 - fy) * g[:, 3:4]
    bw_s, bh_s = torch.where(ok, bw, torch.ones_like(bw)), torch.where(ok, bh, torch.ones_like(bh))
    cx = ((gx - x1.unsqueeze(-1)) / bw_s * map_size).trunc()                   # python int(): toward zero
    cy = ((gy - y1.unsqueeze(-1)) / bh_s * map_size).trunc()
    sub = torch.as_tensor(sub_regions, dtype=torch.float32, device=dev)        # [P, 4]
    xs = torch.arange(half, dtype=torch.float32, device=dev)
    X = (xs[None, :] + sub[:, 0:1])[None, :, None, :]                          # [1, P, 1, half] absolute columns
    Y = (xs[None, :] + sub[:, 1:2])[None, :, :, None]                          # [1, P, half, 1] absolute rows
    d2 = (X - cx[:, :, None, None]) ** 2 + (Y - cy[:, :, None, None]) ** 2
    return ((d2 <= radius * radius) & ok[:, :, None, None]).float()


class GridLossComputation(object):
    def __init__(self, stage, loss_weight, proposal_matcher, pos_radius, grid_points, roi_feat_size):
        self.stage, self.loss_weight, self.proposal_matcher = stage, loss_weight, proposal_matcher
        self.pos_radius, self.grid_points, self.roi_feat_size = pos_radius, grid_points, roi_feat_size
        self.whole_map_size = roi_feat_size * 4
        self.grid_size = int(np.sqrt(grid_points))
        self.sub_regions = calc_sub_regions(grid_points, self.grid_size, self.whole_map_size)

    def subsample(self, proposals, targets):
        if cfg.GRID_RCNN.BETTER_ROI:
            raise ValueError("GRID_RCNN.BETTER_ROI is outside the hot path")
        bboxes, gt_bboxes, new_proposals, quality = [], [], [], []
        for p, t in zip(proposals, targets):
            q = boxlist_iou(t, p)
            matched = self.proposal_matcher(q)
            pos = matched >= 0
            if cfg.GRID_RCNN.IOU_HELPER:
                quality.append(q[:, pos])
            gt = t.bbox[matched.clamp(min=0)]
            if self.stage != 0:
                p, gt = p[pos], gt[pos]
            new_proposals.append(p)
            bboxes.append(p.bbox)
            gt_bboxes.append(gt)
        self.pos_result = (torch.cat(bboxes, dim=0), torch.cat(gt_bboxes, dim=0))
        self.match_quality_matrixs = quality
        return new_proposals

    def prepare_iou_target(self):
        out = []
        for q in self.match_quality_matrixs:
            fg, _ = q.max(dim=0)
            out.append(torch.stack([1 - fg, fg], dim=1))
        return torch.cat(out)

    def prepare_target(self, proposals=None, targets=None):
        pos_bboxes, pos_gt_bboxes = self.pos_result
        assert pos_bboxes.shape == pos_gt_bboxes.shape
        ratio = cfg.GRID_RCNN.CASCADE_MAPPING_OPTION.STAGE_MAPPING_RATIO[self.stage]
        if cfg.GRID_RCNN.TARGET_REFINE:
            raise ValueError("GRID_RCNN.TARGET_REFINE is outside the hot path")
        return grid_targets(pos_bboxes, pos_gt_bboxes, ratio, self.grid_points, self.whole_map_size, self.pos_radius,
                            self.sub_regions)

    def loss_grid(self, grid_logit, grid_target):
        return self.loss_weight * F.binary_cross_entropy_with_logits(grid_logit, grid_target.float())

    def __call__(self, proposals, grid_logits, iou_logits, targets):
        grid_targets_ = self.prepare_target(proposals, targets)
        loss_grid = self.loss_grid(grid_logits["unfused"], grid_targets_)
        G = cfg.GRID_RCNN
        if G.IOU_HELPER and self.stage == G.CASCADE_MAPPING_OPTION.STAGE_NUM - 1:
            loss_iou = l2_loss(iou_logits, self.prepare_iou_target())
        else:
            loss_iou = 0
        return loss_grid, loss_iou


def loss_evaluator(stage=0, type=None):
    G = cfg.GRID_RCNN
    if type == "cls":
        return CLSLossComputation(Matcher(G.FG_IOU_THRESHOLD, G.BG_IOU_THRESHOLD, allow_low_quality_matches=False),
                                  BalancedPositiveNegativeSampler(G.BATCH_SIZE_PER_IMAGE, G.POSITIVE_FRACTION),
                                  cfg.MODEL.CLS_AGNOSTIC_BBOX_REG)
    if type == "grid":
        M = G.CASCADE_MAPPING_OPTION
        points = M.GRID_NUM[stage] if G.CASCADE_MAPPING_ON else G.GRID_POINTS
        return GridLossComputation(stage, G.LOSS_WEIGHT,
                                   Matcher(M.FG_IOU_THRESHOLD[stage], M.BG_IOU_THRESHOLD[stage],
                                           allow_low_quality_matches=False),
                                   G.POS_RADIUS, points, G.ROI_FEAT_SIZE)
    raise Exception("Type error!")
