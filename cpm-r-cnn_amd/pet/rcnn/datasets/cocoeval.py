"""COCO bounding-box evaluation without pycocotools.

The reference scores detections with its own copy of pycocotools' COCOeval (pet/rcnn/datasets/mycocoeval.py:60-480,
driven from evaluation.py:111-136) on top of pycocotools' compiled IoU routine; pycocotools is a third-party dependency
that is absent from /root/reference and from this image (the reference pins no version; its mycocoeval.py is the
published COCOeval of cocoapi with three extra summary rows).  This module restates the PUBLISHED algorithm for
iouType = 'bbox' in numpy:

  per (image, category): detections by descending score (stable), at most maxDets[-1] = 100; IoU of [x, y, w, h] boxes
    = intersection / union, against a crowd ground truth = intersection / detection area;
  evaluateImg: ground truths ignored when `iscrowd` / `ignore` or when their area is outside the area range, sorted
    non-ignored first; every detection, in score order and for each IoU threshold, takes the still-free ground truth
    of highest IoU >= threshold (a crowd may be matched repeatedly; once matched to a regular ground truth a detection
    does not move on to ignored ones); unmatched detections whose own area is outside the range are ignored;
  accumulate: per (category, area range, maxDets) detections of all images merged by descending score (mergesort),
    cumulative tp / fp over the non-ignored ones, precision made monotonically non-increasing from the right and
    sampled at the 101 recall thresholds by searchsorted(side='left'); recall = last tp / number of non-ignored gts;
  summarize: means over the entries that are not -1: AP, AP50, AP75, APs, APm, APl, AR1, AR10, AR100, ARs, ARm, ARl and
    the reference's extra AP60 / AP70 / AP80 / AP90 (mycocoeval.py:476-479; AP90 from the threshold index the
    reference hard-codes there).

Pinned to the reference's vendored evaluator: tests/golden/cocoeval_ref.* is written by running mycocoeval.py:62-423
itself (make_golden.py cocoeval; only pycocotools' box IoU routine, its one third-party call, is supplied by the harness)
and tests/test_cocoeval.py holds this module's precision / recall arrays EQUAL to it and the 16 summary numbers to
1e-12; the known-answer tests stay."""
import collections

import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
REC_THRS = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
MAX_DETS = (1, 10, 100)
AREA_RNG = ((0 ** 2, 1e5 ** 2), (0 ** 2, 32 ** 2), (32 ** 2, 96 ** 2), (96 ** 2, 1e5 ** 2))
AREA_LBL = ("all", "small", "medium", "large")


def box_iou_xywh(dt, gt, iscrowd):
    """[D,4] x [G,4] (x, y, w, h) -> [D,G]; column g with iscrowd[g]: intersection / area(dt)."""
    dt, gt = np.asarray(dt, np.float64).reshape(-1, 4), np.asarray(gt, np.float64).reshape(-1, 4)
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    ix = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    iy = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.clip(ix, 0, None) * np.clip(iy, 0, None)
    ad, ag = dt[:, 2] * dt[:, 3], gt[:, 2] * gt[:, 3]
    union = np.where(np.asarray(iscrowd, bool)[None, :], ad[:, None], ad[:, None] + ag[None, :] - inter)
    return inter / np.maximum(union, np.finfo(np.float64).tiny)


class COCOBoxEval(object):
    """gt: the dataset's COCO json dict ('images', 'annotations', 'categories'); dt: list of result records
    {'image_id', 'category_id', 'bbox': [x, y, w, h], 'score'}."""

    def __init__(self, gt, dt):
        self.img_ids = sorted(im["id"] for im in gt["images"])
        self.cat_ids = sorted(c["id"] for c in gt["categories"])
        self.gts, self.dts = collections.defaultdict(list), collections.defaultdict(list)
        for a in gt["annotations"]:
            g = dict(a)
            g["ignore"] = int(g.get("ignore", 0) or g.get("iscrowd", 0))
            g.setdefault("area", g["bbox"][2] * g["bbox"][3])
            self.gts[g["image_id"], g["category_id"]].append(g)
        for i, d in enumerate(dt):
            d = dict(d)
            d["area"] = d["bbox"][2] * d["bbox"][3]
            d["id"] = i + 1
            self.dts[d["image_id"], d["category_id"]].append(d)
        self.eval_imgs, self.eval, self.stats = None, None, None

    def _evaluate_img(self, img, cat, ious, rng, max_det):
        gt, dt = self.gts.get((img, cat), []), self.dts.get((img, cat), [])
        if not gt and not dt:
            return None
        g_ig = np.array([g["ignore"] or g["area"] < rng[0] or g["area"] > rng[1] for g in gt], bool)
        gorder = np.argsort(g_ig, kind="mergesort")
        gt = [gt[i] for i in gorder]
        g_ig = g_ig[gorder]
        dorder = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in dorder[:max_det]]
        iscrowd = np.array([int(g.get("iscrowd", 0)) for g in gt], bool)
        iou = ious[:, gorder][:len(dt)] if len(ious) else ious
        T, G, D = len(IOU_THRS), len(gt), len(dt)
        gtm, dtm, d_ig = np.zeros((T, G)), np.zeros((T, D)), np.zeros((T, D), bool)
        if G and D:
            for ti, t in enumerate(IOU_THRS):
                for di in range(D):
                    best, m = min(t, 1 - 1e-10), -1
                    for gi in range(G):
                        if gtm[ti, gi] > 0 and not iscrowd[gi]:
                            continue
                        if m > -1 and not g_ig[m] and g_ig[gi]:
                            break
                        if iou[di, gi] < best:
                            continue
                        best, m = iou[di, gi], gi
                    if m == -1:
                        continue
                    d_ig[ti, di] = g_ig[m]
                    dtm[ti, di] = gt[m]["id"]
                    gtm[ti, m] = dt[di]["id"]
        out = np.array([d["area"] < rng[0] or d["area"] > rng[1] for d in dt], bool).reshape(1, D)
        d_ig = d_ig | ((dtm == 0) & np.repeat(out, T, 0))
        return dict(dtm=dtm, d_ig=d_ig, g_ig=g_ig, scores=np.array([d["score"] for d in dt]))

    def evaluate(self):
        ious = {}
        for img in self.img_ids:
            for cat in self.cat_ids:
                gt, dt = self.gts.get((img, cat), []), self.dts.get((img, cat), [])
                order = np.argsort([-d["score"] for d in dt], kind="mergesort")[:MAX_DETS[-1]]
                ious[img, cat] = box_iou_xywh([dt[i]["bbox"] for i in order], [g["bbox"] for g in gt],
                                              [int(g.get("iscrowd", 0)) for g in gt])
        self.eval_imgs = {(cat, ai, img): self._evaluate_img(img, cat, ious[img, cat], rng, MAX_DETS[-1])
                          for cat in self.cat_ids for ai, rng in enumerate(AREA_RNG) for img in self.img_ids}
        return self

    def accumulate(self):
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(self.cat_ids), len(AREA_RNG), len(MAX_DETS)
        precision, recall = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M))
        for k, cat in enumerate(self.cat_ids):
            for a in range(A):
                E = [self.eval_imgs[cat, a, img] for img in self.img_ids]
                E = [e for e in E if e is not None]
                if not E:
                    continue
                for m, max_det in enumerate(MAX_DETS):
                    scores = np.concatenate([e["scores"][:max_det] for e in E])
                    inds = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([e["dtm"][:, :max_det] for e in E], axis=1)[:, inds]
                    d_ig = np.concatenate([e["d_ig"][:, :max_det] for e in E], axis=1)[:, inds]
                    npig = int(np.count_nonzero(~np.concatenate([e["g_ig"] for e in E])))
                    if npig == 0:
                        continue
                    tp_sum = np.cumsum((dtm != 0) & ~d_ig, axis=1).astype(np.float64)
                    fp_sum = np.cumsum((dtm == 0) & ~d_ig, axis=1).astype(np.float64)
                    for t in range(T):
                        tp, fp = tp_sum[t], fp_sum[t]
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        q = np.zeros(R)
                        pos = np.searchsorted(rc, REC_THRS, side="left")
                        for ri, pi in enumerate(pos):
                            if pi < nd:
                                q[ri] = pr[pi]
                        precision[t, :, k, a, m] = q
        self.eval = dict(precision=precision, recall=recall)
        return self

    def _summ(self, ap, iou_thr=None, area="all", max_det=100):
        a, m = AREA_LBL.index(area), MAX_DETS.index(max_det)
        s = self.eval["precision"] if ap else self.eval["recall"]
        if iou_thr is not None:
            s = s[np.where(np.isclose(IOU_THRS, iou_thr))[0]]
        s = s[..., a, m]
        s = s[s > -1]
        return float(np.mean(s)) if s.size else -1.0

    def summarize(self):
        names = ["AP", "AP50", "AP75", "APs", "APm", "APl", "AR1", "AR10", "AR100", "ARs", "ARm", "ARl", "AP60", "AP70",
                 "AP80", "AP90"]
        vals = [self._summ(1), self._summ(1, 0.5), self._summ(1, 0.75), self._summ(1, area="small"),
                self._summ(1, area="medium"), self._summ(1, area="large"), self._summ(0, max_det=1),
                self._summ(0, max_det=10), self._summ(0, max_det=100), self._summ(0, area="small"),
                self._summ(0, area="medium"), self._summ(0, area="large"), self._summ(1, 0.6), self._summ(1, 0.7),
                self._summ(1, 0.8), self._summ(1, 0.9)]
        self.stats = collections.OrderedDict(zip(names, vals))
        return self.stats


def evaluate_boxes(gt_json, detections):
    return COCOBoxEval(gt_json, detections).evaluate().accumulate().summarize()
