"""BoxList helpers of the detection path (counterpart of pet/utils/data/structures/boxlist_ops.py):
boxlist_nms (:10-32), boxlist_ml_nms (:35-61), remove_small_boxes (:104-118), boxlist_iou with the
"+1" convention (:123-158), cat_boxlist (:172-198)."""
import torch

from pet.lib.ops import nms as _box_nms
from pet.lib.ops import ml_nms as _box_ml_nms
from pet.utils.data.structures.bounding_box import BoxList


def boxlist_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores"):
    if nms_thresh <= 0:
        return boxlist
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = _box_nms(boxlist.bbox, boxlist.get_field(score_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep].convert(mode)


def boxlist_ml_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores", label_field="labels"):
    if nms_thresh <= 0:
        return boxlist
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = _box_ml_nms(boxlist.bbox, boxlist.get_field(score_field), boxlist.get_field(label_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep].convert(mode)


def remove_small_boxes(boxlist, min_size):
    wh = boxlist.convert("xywh").bbox
    keep = ((wh[:, 2] >= min_size) & (wh[:, 3] >= min_size)).nonzero().squeeze(1)
    return boxlist[keep]


def box_iou_plus1(box1, box2):
    """[N,4] x [M,4] -> [N,M] IoU with pixel-inclusive (+1) widths; same operation order as the reference."""
    area1 = (box1[:, 2] - box1[:, 0] + 1) * (box1[:, 3] - box1[:, 1] + 1)
    area2 = (box2[:, 2] - box2[:, 0] + 1) * (box2[:, 3] - box2[:, 1] + 1)
    lt = torch.max(box1[:, None, :2], box2[:, :2])
    rb = torch.min(box1[:, None, 2:], box2[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2 - inter)


def boxlist_iou(boxlist1, boxlist2):
    if boxlist1.size != boxlist2.size:
        raise RuntimeError("boxlists should have same image size, got {}, {}".format(boxlist1, boxlist2))
    return box_iou_plus1(boxlist1.convert("xyxy").bbox, boxlist2.convert("xyxy").bbox)


def _cat(tensors, dim=0):
    assert isinstance(tensors, (list, tuple))
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)


def cat_boxlist(bboxes):
    assert isinstance(bboxes, (list, tuple)) and all(isinstance(b, BoxList) for b in bboxes)
    size, mode, fields = bboxes[0].size, bboxes[0].mode, set(bboxes[0].fields())
    assert all(b.size == size and b.mode == mode and set(b.fields()) == fields for b in bboxes)
    out = BoxList(_cat([b.bbox for b in bboxes], dim=0), size, mode)
    for f in fields:
        out.add_field(f, _cat([b.get_field(f) for b in bboxes], dim=0))
    return out
