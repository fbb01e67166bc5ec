"""Test-time post-processing of the offset-regression heads (counterpart of pet/rcnn/core/test.py:433-509
`filter_results`): score threshold, per-class (soft-)NMS, optional box voting, cap on detections per image.  The
reference loops over the 80 classes with one NMS call -- and for soft-NMS one device->host->device round trip -- each;
here all classes go through ONE batched device launch per step (cpm_nms_batched / cpm_soft_nms_batched).

Only the post-processing lives here: the reference's image driver around it (im_detect_bbox / get_blob: cv2 resize,
flip / multi-scale test-time augmentation) depends on OpenCV and is not part of this tree."""
import numpy as np
import torch

from pet.lib.ops import soft_nms_segments
from pet.lib.ops.boxlist_ops import boxlist_box_voting, boxlist_ml_nms, boxlist_nms
from pet.lib.ops.nms import SOFT_NMS_METHODS
from pet.rcnn.core.config import cfg
from pet.utils.data.structures.bounding_box import BoxList
from pet.utils.data.structures.boxlist_ops import cat_boxlist


def _limit(result):
    """Keep the DETECTIONS_PER_IMG best over all classes (test.py:500-508: kthvalue threshold, ties kept)."""
    n = len(result)
    if n > cfg.FAST_RCNN.DETECTIONS_PER_IMG > 0:
        scores = result.get_field("scores")
        thresh, _ = torch.kthvalue(scores.cpu(), n - cfg.FAST_RCNN.DETECTIONS_PER_IMG + 1)
        result = result[torch.nonzero(scores >= thresh.item()).squeeze(1)]
    return result


def filter_results(boxlist):
    """boxlist: [R * num_classes] boxes (class-major per RoI) with a 'scores' field; with soft-NMS / voting it also
    carries 'labels' (test.py:448-450)."""
    num_classes = cfg.MODEL.NUM_CLASSES
    T = cfg.TEST
    if not T.SOFT_NMS.ENABLED and not T.BBOX_VOTE.ENABLED:
        scores = boxlist.get_field("scores")
        n = boxlist.bbox.shape[0]
        labels = torch.arange(n, device=scores.device) % num_classes
        boxlist.add_field("labels", labels.to(torch.int64))
        keep = (scores > cfg.FAST_RCNN.SCORE_THRESH) & (labels != 0)
        return _limit(boxlist_ml_nms(boxlist[keep], cfg.FAST_RCNN.NMS))
    boxes = boxlist.bbox.reshape(-1, 4)
    labels = boxlist.get_field("labels")
    scores = boxlist.get_field("scores")
    device = scores.device
    ok = scores > cfg.FAST_RCNN.SCORE_THRESH
    # group the candidates by class once (one host round trip: the per-class counts)
    cand = torch.nonzero(ok & (labels >= 1) & (labels < num_classes)).squeeze(1)
    order = cand[torch.sort(labels[cand], stable=True)[1]]
    counts = torch.bincount(labels[order], minlength=num_classes).cpu().numpy()
    offsets = np.concatenate([[0], np.cumsum(counts)])
    cboxes, cscores = boxes[order].contiguous(), scores[order].contiguous()
    per_class = {}
    if T.SOFT_NMS.ENABLED and cfg.FAST_RCNN.NMS > 0:
        method = SOFT_NMS_METHODS[T.SOFT_NMS.METHOD]
        for lo in range(1, num_classes, 64):                          # <= 64 segments per launch
            hi = min(lo + 64, num_classes)
            seg = (offsets[lo:hi + 1] - offsets[lo]).tolist()
            a, z = int(offsets[lo]), int(offsets[hi])
            if z == a:
                continue
            b, s, _, c = soft_nms_segments(cboxes[a:z], cscores[a:z], seg, T.SOFT_NMS.SIGMA, cfg.FAST_RCNN.NMS, 0.0001,
                                           method)
            c = c.cpu().numpy()
            for j in range(lo, hi):
                o = seg[j - lo]
                per_class[j] = (b[o:o + c[j - lo]], s[o:o + c[j - lo]])
    result = []
    for j in range(1, num_classes):
        a, z = int(offsets[j]), int(offsets[j + 1])
        old = BoxList(cboxes[a:z], boxlist.size, mode="xyxy")
        old.add_field("scores", cscores[a:z])
        if j in per_class:
            cur = BoxList(per_class[j][0], boxlist.size, mode="xyxy")
            cur.add_field("scores", per_class[j][1])
        elif T.SOFT_NMS.ENABLED:
            cur = old
        else:
            cur = boxlist_nms(old, cfg.FAST_RCNN.NMS)
        if T.BBOX_VOTE.ENABLED and z > a:
            cur = boxlist_box_voting(cur, old, T.BBOX_VOTE.VOTE_TH, scoring_method=T.BBOX_VOTE.SCORING_METHOD)
        cur.add_field("labels", torch.full((len(cur),), j, dtype=torch.int64, device=device))
        result.append(cur)
    return _limit(cat_boxlist(result))
