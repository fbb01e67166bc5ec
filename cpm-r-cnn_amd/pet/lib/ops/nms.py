"""NMS operator surface (reference: pet/lib/ops/nms.py:10-13).  fp32-only, device-only."""
from . import _C


def nms(boxes, scores, iou_threshold):
    return _C.nms(boxes.float(), scores.float(), float(iou_threshold))


def ml_nms(boxes, scores, labels, iou_threshold, topk=0):
    return _C.ml_nms(boxes.float(), scores.float(), labels, float(iou_threshold), int(topk))


nms_segments = _C.nms_segments

SOFT_NMS_METHODS = {"hard": 0, "linear": 1, "gaussian": 2}          # nms.py:5


def soft_nms(dets, scores, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method="linear"):
    """Soft-NMS (https://arxiv.org/abs/1704.04503) on the device; reference: nms.py:16-28 (a CPU kernel there)."""
    assert method in SOFT_NMS_METHODS, "Unknown soft_nms method: {}".format(method)
    return _C.soft_nms(dets, scores, sigma, overlap_thresh, score_thresh, SOFT_NMS_METHODS[method])


def ml_soft_nms(dets, scores, labels, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method="linear", topk=0):
    """Multi-label soft-NMS on the device; reference: nms.py:31-45.  NB the reference kernel stops when `topk == i`, so
    its default topk = 0 keeps nothing; pass a negative topk for "no limit"."""
    assert method in SOFT_NMS_METHODS, "Unknown soft_nms method: {}".format(method)
    return _C.ml_soft_nms(dets, scores, labels, sigma, overlap_thresh, score_thresh, SOFT_NMS_METHODS[method], topk)


soft_nms_segments = _C.soft_nms_segments
