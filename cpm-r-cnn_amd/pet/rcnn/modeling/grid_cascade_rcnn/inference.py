"""Post-processing of the CPM head (counterpart of pet/rcnn/modeling/grid_cascade_rcnn/inference.py:32-320).

CLSPostProcessor: softmax -> per-class candidates (score > thresh, fg) -> multi-label NMS (device), and the RSM
re-scoring s^0.8 * p^0.2.  GridPostProcessor: heat maps -> boxes (per-point arg-max in its 28x28 window, map to
image coordinates through the stage's mapping ratio, score-weighted vote of the 3 points on each side).  The
reference runs get_boxes on the CPU (.cpu() :195-196, .cuda() :278); here it stays on the device.
Reference quirks kept on purpose: decoded boxes are NOT clipped (the clamp_ at :275-276 acts on a
copy); test-time ISM/RSM use whole-batch tensors, i.e. one image per forward."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from pet.lib.ops.boxlist_ops import boxlist_ml_nms
from pet.rcnn.core.config import cfg
from pet.rcnn.modeling.grid_rcnn.loss import calc_sub_regions
from pet.utils.data.structures.bounding_box import BoxList
from pet.utils.data.structures.boxlist_ops import cat_boxlist


class CLSPostProcessor(nn.Module):
    def __init__(self, score_thresh, nms):
        super().__init__()
        self.score_thresh = score_thresh
        self.nms = nms

    def forward(self, x, boxes, rescore=False):
        class_prob = F.softmax(x, -1)
        if rescore:
            for b in boxes:       # whole-batch arange, as inference.py:65 (one image per forward)
                p = class_prob[torch.arange(class_prob.shape[0], device=class_prob.device), b.get_field("labels")]
                b.add_field("scores", (b.get_field("scores") ** 0.8) * (p ** 0.2))
            return boxes
        image_shapes = [b.size for b in boxes]
        per_image = [len(b) for b in boxes]
        concat = torch.cat([b.bbox for b in boxes], dim=0)
        num_classes = class_prob.shape[1]
        props = concat.repeat(1, num_classes).split(per_image, dim=0)
        probs = class_prob.split(per_image, dim=0)
        out = []
        for prob, bx, shape in zip(probs, props, image_shapes):
            bl = BoxList(bx.reshape(-1, 4), shape, mode="xyxy")
            bl.add_field("scores", prob.reshape(-1))
            bl = bl.clip_to_image(remove_empty=False)
            out.append(self.filter_results(bl, num_classes))
        return out

    def filter_results(self, boxlist, num_classes):
        scores = boxlist.get_field("scores")
        n = boxlist.bbox.shape[0]
        labels = torch.arange(n, device=scores.device) % num_classes
        boxlist.add_field("labels", labels.to(torch.int64))
        keep = (scores > self.score_thresh) & (labels != 0)
        return boxlist_ml_nms(boxlist[keep], self.nms)


_DECODE_CONSTS = {}


def first_argmax(flat):
    """max over dim 1 and the FIRST index attaining it (deterministic on the GPU)."""
    mx = flat.max(dim=1, keepdim=True)[0]
    n = flat.shape[1]
    idx = torch.where(flat == mx, torch.arange(n, device=flat.device)[None], torch.full((1, 1), n, device=flat.device))
    return mx.squeeze(1), idx.min(dim=1)[0]


def decode_grid_boxes(det_bboxes, grid_pred, mapping_ratio, grid_points, sub_regions, whole_map_size):
    """[R,4] RoIs + [R,P,h,w] logits -> [R,4] refined boxes (inference.py:189-279)."""
    R, c, h, w = grid_pred.shape
    gs = int(np.sqrt(grid_points))
    half = whole_map_size // 4 * 2
    assert h == w == half and c == grid_points
    prob = grid_pred.sigmoid().reshape(R * c, h * w)
    pred_scores, pos = first_argmax(prob)
    # the window origins and the four sides' point indices live on the device (built once per geometry): a Python list
    # used as an index or handed to as_tensor is a blocking host-to-device copy per use -- seven per stage, 0.6 ms each
    # behind a busy stream (the test-time forward spent 4 of its 11 ms there)
    key = (grid_pred.device, grid_points, tuple(map(tuple, sub_regions)) if not torch.is_tensor(sub_regions) else id(sub_regions))
    cached = _DECODE_CONSTS.get(key)
    if cached is None:
        dev = grid_pred.device
        cached = _DECODE_CONSTS[key] = (
            torch.as_tensor(sub_regions, dtype=torch.int64, device=dev),
            torch.as_tensor(list(range(gs)), dtype=torch.int64, device=dev),
            torch.as_tensor([i * gs for i in range(gs)], dtype=torch.int64, device=dev),
            torch.as_tensor([grid_points - gs + i for i in range(gs)], dtype=torch.int64, device=dev),
            torch.as_tensor([(i + 1) * gs - 1 for i in range(gs)], dtype=torch.int64, device=dev))
    sub, x1_inds, y1_inds, x2_inds, y2_inds = cached
    xs = (pos % w).view(R, c) + sub[:, 0][None]
    ys = torch.div(pos, w, rounding_mode="floor").view(R, c) + sub[:, 1][None]
    pred_scores = pred_scores.view(R, c)
    widths = (det_bboxes[:, 2] - det_bboxes[:, 0]).unsqueeze(-1)
    heights = (det_bboxes[:, 3] - det_bboxes[:, 1]).unsqueeze(-1)
    x1 = det_bboxes[:, 0, None] - mapping_ratio * (widths / 2)
    y1 = det_bboxes[:, 1, None] - mapping_ratio * (heights / 2)
    abs_xs = (xs.float() + 0.5) / (2 * w) * (1 + mapping_ratio) * widths + x1
    abs_ys = (ys.float() + 0.5) / (2 * h) * (1 + mapping_ratio) * heights + y1

    def vote(coord, inds):
        s = pred_scores[:, inds]
        return (coord[:, inds] * s).sum(dim=1, keepdim=True) / s.sum(dim=1, keepdim=True)

    return torch.cat([vote(abs_xs, x1_inds), vote(abs_ys, y1_inds), vote(abs_xs, x2_inds), vote(abs_ys, y2_inds)],
                     dim=1)


class GridPostProcessor(nn.Module):
    def __init__(self, stage, grid_points, roi_feat_size, nms_on=True):
        super().__init__()
        self.stage, self.grid_points, self.roi_feat_size = stage, grid_points, roi_feat_size
        self.whole_map_size = roi_feat_size * 4
        self.grid_size = int(np.sqrt(grid_points))
        self.sub_regions = calc_sub_regions(grid_points, self.grid_size, self.whole_map_size)
        self.nms_on = nms_on

    def forward(self, grid_logits, proposals, iou_logits=None, is_train=False, targets=None):
        G = cfg.GRID_RCNN
        grid_pred = grid_logits["fused"] if G.FUSED_ON else grid_logits["unfused"]
        if G.CASCADE_MAPPING_OPTION.RESIZE_ROI:
            raise ValueError("RESIZE_ROI is outside the hot path")
        last = G.IOU_HELPER and self.stage == G.CASCADE_MAPPING_OPTION.STAGE_NUM - 1
        out = []
        for i, p in enumerate(proposals):
            n = p.bbox.shape[0]
            pred, grid_pred = grid_pred[:n], grid_pred[n:]
            if is_train:
                keep = self._filter_boxes(p, targets[i])
                p = p[keep]
                if p.bbox.shape[0] != 0:
                    p.bbox = self.get_boxes(p, pred[keep], is_train)
                p = self.add_gt_proposals(p, targets[i])
            else:
                box = self.get_boxes(p, pred, is_train)
                if last:
                    score, iou_score = p.get_field("scores"), iou_logits[:, 1]
                    assert score.shape == iou_score.shape
                    p.add_field("scores", score * iou_score if G.IOU_HELPER_MERGE else iou_score)
                p.bbox = box
            out.append(p)
        return out

    def get_boxes(self, proposals, grid_pred, is_train):
        det = proposals.bbox
        assert det.shape[0] > 0 and det.shape[0] == grid_pred.shape[0]
        ratio = 1 if cfg.GRID_RCNN.EXTEND_ROI else cfg.GRID_RCNN.CASCADE_MAPPING_OPTION.STAGE_MAPPING_RATIO[self.stage]
        return decode_grid_boxes(det, grid_pred, ratio, self.grid_points, self.sub_regions, self.whole_map_size)

    def _filter_boxes(self, proposal, target):
        """Drop RoIs whose coordinates coincide with a gt (inference.py:281-290): every coordinate equal to the same
        coordinate of ANY gt is replaced by -1, the RoI survives while the 4 replaced coordinates sum to > 0."""
        b, gt = proposal.bbox, target.bbox
        hit = (b[:, None, :] == gt[None, :, :]).any(dim=1)
        r = torch.where(hit, torch.full_like(b, -1), b)
        return torch.nonzero(((r[:, 0] + r[:, 1]) + r[:, 2]) + r[:, 3] > 0).squeeze(1)

    def add_gt_proposals(self, proposal, target):
        gt = target.copy_with_fields(["labels"])
        gt.add_field("objectness", torch.ones(len(gt), device=proposal.bbox.device))
        return cat_boxlist((proposal, gt))


def post_processor(stage=0, type=None):
    G = cfg.GRID_RCNN
    if type == "cls":
        return CLSPostProcessor(G.SCORE_THRESH, G.NMS)
    if type == "grid":
        points = G.CASCADE_MAPPING_OPTION.GRID_NUM[stage] if G.CASCADE_MAPPING_ON else G.GRID_POINTS
        return GridPostProcessor(stage, points, G.ROI_FEAT_SIZE)
    raise Exception("Type error!")
