"""Test-time post-processing of the offset-regression heads (counterpart of pet/rcnn/core/test.py:433-509
`filter_results`): score threshold, per-class (soft-)NMS, optional box voting, cap on detections per image.  The
reference loops over the 80 classes with one NMS call -- and for soft-NMS one device->host->device round trip -- each;
here all classes go through ONE batched device launch per step (cpm_nms_batched / cpm_soft_nms_batched).

The image driver (test.py:12-47,299-358): get_blob resizes on the device (cpm_image_resize_linear = cv2.resize INTER_LINEAR
on float data, the reference's call), im_detect_bbox runs box_net once, plus once per flip / scale of
TEST.BBOX_AUG, maps every result back to the first pass' image frame and concatenates."""
import numpy as np
import torch

from pet.lib.ops import soft_nms_segments
from pet.lib.ops.boxlist_ops import boxlist_box_voting, boxlist_ml_nms, boxlist_nms
from pet.lib.ops.nms import SOFT_NMS_METHODS
from pet.rcnn.core.config import cfg
from pet.utils.data.structures.bounding_box import BoxList
from pet.utils.data.structures.boxlist_ops import cat_boxlist


def get_blob(ims, target_scale, target_max_size, flip):
    """ims: decoded images, uint8 [H,W,3] RGB (numpy or tensor).  Returns fp32 BGR [3,h,w] device tensors, the shorter
    side scaled to target_scale unless the longer would exceed target_max_size (test.py:340-358)."""
    from pet.lib.ops import resize_linear
    device = torch.device(cfg.DEVICE)
    out = []
    for im in ims:
        t = torch.from_numpy(np.array(im, dtype=np.uint8, order="C")) if not torch.is_tensor(im) else im
        size_min, size_max = min(t.shape[0], t.shape[1]), max(t.shape[0], t.shape[1])
        scale = float(target_scale) / float(size_min)
        if np.round(scale * size_max) > target_max_size:
            scale = float(target_max_size) / float(size_max)
        out.append(resize_linear(t.to(device), scale, flip=flip, swap_rb=True))
    return out


def im_detect_bbox_net(model, ims, target_scale, target_max_size, flip=False, size=None):
    """One pass of box_net; flipped results are mirrored back and, with `size`, resized to the first pass' frame
    (test.py:299-327)."""
    blob = get_blob(ims, target_scale, target_max_size, flip)
    results, net_sizes = [], []
    with torch.no_grad():
        feats, raw = model.box_net(blob)
        for i, r in enumerate(raw):
            net_sizes.append(r.size)
            if flip:
                r = r.transpose(0)
                if len(cfg.TRAIN.LEFT_RIGHT) > 0:
                    raise NotImplementedError("left/right class swapping is outside the COCO box path")
            if size:
                r = r.resize(size[i])
            results.append(r)
    return results, net_sizes, feats


def im_detect_bbox(model, ims):
    """(test.py:12-47)"""
    A = cfg.TEST.BBOX_AUG
    per_image = [[] for _ in ims]
    features = []

    def add(res, sizes, feats):
        for i, r in enumerate(res):
            per_image[i].append(r)
        features.append((sizes, feats))
    res, net_sizes, feats = im_detect_bbox_net(model, ims, cfg.TEST.SCALE, cfg.TEST.MAX_SIZE)
    if cfg.MODEL.RPN_ONLY:
        return res, None
    add(res, net_sizes, feats)
    if A.ENABLED:
        if A.H_FLIP:
            add(*im_detect_bbox_net(model, ims, cfg.TEST.SCALE, cfg.TEST.MAX_SIZE, True, net_sizes))
        for scale in A.SCALES:
            add(*im_detect_bbox_net(model, ims, scale, A.MAX_SIZE, False, net_sizes))
            if A.H_FLIP:
                add(*im_detect_bbox_net(model, ims, scale, A.MAX_SIZE, True, net_sizes))
    results = [cat_boxlist(r) for r in per_image]
    if not cfg.MODEL.GRID_ON or A.ENABLED:
        results = [filter_results(r) for r in results]
    return results, features


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
    scores = boxlist.get_field("scores")
    device = scores.device
    if boxlist.has_field("labels"):
        labels = boxlist.get_field("labels")
    else:
        # class-major layout of the offset-regression heads (one box per class and RoI); the reference reads a "labels"
        # field here (test.py:449), which only the grid results carry -- without it the class is the position
        labels = torch.arange(boxes.shape[0], device=device) % num_classes
    ok = scores > cfg.FAST_RCNN.SCORE_THRESH
    # group the candidates by class once (one host round trip: the per-class counts)
    cand = torch.nonzero(ok & (labels >= 1) & (labels < num_classes)).squeeze(1)
    order = cand[torch.sort(labels[cand], stable=True)[1]]
    counts = torch.bincount(labels[order], minlength=num_classes).cpu().numpy()
    offsets = np.concatenate([[0], np.cumsum(counts)])
    cboxes, cscores = boxes[order].contiguous(), scores[order].contiguous()
    per_class = {}
    if T.SOFT_NMS.ENABLED:                   # whenever enabled, whatever FAST_RCNN.NMS is (reference test.py:468-476)
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
