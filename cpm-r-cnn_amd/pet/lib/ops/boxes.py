"""Dense box IoU (counterpart of pet/lib/ops/boxes.py:46-47 -> _C.box_iou), areas without +1."""
from . import _C


def box_iou(boxes, query_boxes):
    return _C.box_iou(boxes.float(), query_boxes.float())
