"""BoxList NMS helpers of pet/lib/ops/boxlist_ops.py:15-67,289-315 (TO_REMOVE = 0 conventions)."""
import torch

from pet.lib.ops.nms import nms as _box_nms, ml_nms as _box_ml_nms, soft_nms as _box_soft_nms
from pet.utils.data.structures.boxlist_ops import cat_boxlist  # noqa: F401  (re-export, boxlist_ops.py:289)


def boxlist_nms(boxlist, nms_thresh, topk=0, score_field="scores", idxs=None):
    if nms_thresh <= 0:
        return boxlist
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    boxes = boxlist.bbox
    if idxs is not None:      # batched NMS by per-class coordinate offset (boxlist_ops.py:34-38)
        offsets = idxs.to(boxes) * (boxes.max() + torch.tensor(1).to(boxes))
        boxes = boxes + offsets[:, None]
    keep = _box_nms(boxes, boxlist.get_field(score_field), nms_thresh)
    if keep.size(0) > topk > 0:
        keep = keep[:topk]
    return boxlist[keep].convert(mode)


def boxlist_ml_nms(boxlist, nms_thresh, topk=0, score_field="scores", label_field="labels"):
    if nms_thresh <= 0:
        return boxlist
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = _box_ml_nms(boxlist.bbox, boxlist.get_field(score_field), boxlist.get_field(label_field), nms_thresh, topk)
    return boxlist[keep].convert(mode)


def boxlist_soft_nms(boxlist, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method="linear", score_field="scores"):
    """boxlist_ops.py:70-91, without the reference's device -> host -> device round trip."""
    if overlap_thresh <= 0:
        return boxlist
    from pet.utils.data.structures.bounding_box import BoxList
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    dets, scores, _ = _box_soft_nms(boxlist.bbox, boxlist.get_field(score_field), sigma, overlap_thresh, score_thresh,
                                    method)
    out = BoxList(dets, boxlist.size, mode="xyxy")
    out.add_field("scores", scores)
    return out.convert(mode)


def boxlist_box_voting(top_boxlist, all_boxlist, thresh, scoring_method="ID", beta=1.0, score_field="scores"):
    """boxlist_ops.py:120-131."""
    if thresh <= 0:
        return top_boxlist
    from pet.lib.ops.boxes import box_voting
    from pet.utils.data.structures.bounding_box import BoxList
    mode = top_boxlist.mode
    boxes, scores = box_voting(top_boxlist.convert("xyxy").bbox, top_boxlist.get_field(score_field),
                               all_boxlist.convert("xyxy").bbox, all_boxlist.get_field(score_field), thresh,
                               scoring_method, beta)
    out = BoxList(boxes, all_boxlist.size, mode="xyxy")
    out.add_field("scores", scores)
    return out.convert(mode)


def boxlist_ml_soft_nms(boxlist, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method="linear", topk=0,
                        score_field="scores"):
    """boxlist_ops.py:94-117 (multi-label), on the device."""
    if overlap_thresh <= 0:
        return boxlist
    from pet.lib.ops.nms import ml_soft_nms
    from pet.utils.data.structures.bounding_box import BoxList
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    dets, scores, labels, _ = ml_soft_nms(boxlist.bbox, boxlist.get_field(score_field), boxlist.get_field("labels"), sigma,
                                          overlap_thresh, score_thresh, method, topk)
    out = BoxList(dets, boxlist.size, mode="xyxy")
    out.add_field("scores", scores)
    out.add_field("labels", labels)
    return out.convert(mode)


def boxlist_box_ml_voting(top_boxlist, all_boxlist, thresh, scoring_method="ID", beta=1.0, score_field="scores"):
    """boxlist_ops.py:134-148."""
    if thresh <= 0:
        return top_boxlist
    from pet.lib.ops.boxes import box_ml_voting
    from pet.utils.data.structures.bounding_box import BoxList
    mode = top_boxlist.mode
    boxes, scores, labels = box_ml_voting(top_boxlist.convert("xyxy").bbox, top_boxlist.get_field(score_field),
                                          top_boxlist.get_field("labels"), all_boxlist.convert("xyxy").bbox,
                                          all_boxlist.get_field(score_field), all_boxlist.get_field("labels"), thresh,
                                          scoring_method, beta)
    out = BoxList(boxes, all_boxlist.size, mode="xyxy")
    out.add_field("scores", scores)
    out.add_field("labels", labels)
    return out.convert(mode)
