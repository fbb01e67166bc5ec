"""NMS operator surface (reference: pet/lib/ops/nms.py:10-13).  fp32-only, device-only."""
from . import _C


def nms(boxes, scores, iou_threshold):
    return _C.nms(boxes.float(), scores.float(), float(iou_threshold))


def ml_nms(boxes, scores, labels, iou_threshold, topk=0):
    return _C.ml_nms(boxes.float(), scores.float(), labels, float(iou_threshold), int(topk))


nms_segments = _C.nms_segments
