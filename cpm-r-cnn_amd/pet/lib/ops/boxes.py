"""Dense box IoU (counterpart of pet/lib/ops/boxes.py:46-47 -> _C.box_iou), areas without +1."""
from . import _C


def box_iou(boxes, query_boxes):
    return _C.box_iou(boxes.float(), query_boxes.float())

BOX_VOTING_METHODS = {"ID": 0, "TEMP_AVG": 1, "AVG": 2, "IOU_AVG": 3, "GENERALIZED_AVG": 4, "QUASI_SUM": 5}   # boxes.py:3


def box_voting(top_boxes, top_scores, all_boxes, all_scores, overlap_thresh, method="ID", beta=1.0):
    """Refine the kept detections by voting with all detections of the class (https://arxiv.org/abs/1505.01749);
    reference: boxes.py:6-22."""
    assert method in BOX_VOTING_METHODS, "Unknown box_voting method: {}".format(method)
    return _C.box_voting(top_boxes, top_scores, all_boxes, all_scores, BOX_VOTING_METHODS[method], beta, overlap_thresh)


def box_ml_voting(top_boxes, top_scores, top_labels, all_boxes, all_scores, all_labels, overlap_thresh, method="ID",
                  beta=1.0):
    """Multi-label voting: only detections of the same label vote (reference: boxes.py:25-45)."""
    assert method in BOX_VOTING_METHODS, "Unknown box_voting method: {}".format(method)
    return _C.box_ml_voting(top_boxes, top_scores, top_labels, all_boxes, all_scores, all_labels,
                            BOX_VOTING_METHODS[method], beta, overlap_thresh)
