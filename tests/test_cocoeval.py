"""COCO box evaluation (pet/rcnn/datasets/cocoeval.py) -- the restated published COCOeval algorithm, held by known
answers (pycocotools itself is absent from the reference tree and from this image: parity unpinned)."""
import numpy as np

import pytest


def _ds(images, anns, cats=(1, 2)):
    return {"images": [{"id": i, "width": 640, "height": 480} for i in images],
            "categories": [{"id": c, "name": "c%d" % c} for c in cats],
            "annotations": [dict(id=k + 1, image_id=a[0], category_id=a[1], bbox=list(a[2]), iscrowd=a[3] if len(a) > 3 else 0,
                                 area=a[2][2] * a[2][3]) for k, a in enumerate(anns)]}


def _det(img, cat, box, score):
    return {"image_id": img, "category_id": cat, "bbox": list(box), "score": score}


def test_iou_and_crowd_rule():
    from pet.rcnn.datasets.cocoeval import box_iou_xywh
    iou = box_iou_xywh([[0, 0, 10, 10]], [[5, 0, 10, 10], [0, 0, 100, 100]], [0, 1])
    assert abs(iou[0, 0] - 50.0 / 150.0) < 1e-12
    assert abs(iou[0, 1] - 1.0) < 1e-12          # crowd: intersection / detection area


def test_perfect_detections_score_one():
    from pet.rcnn.datasets.cocoeval import evaluate_boxes
    anns = [(1, 1, (10, 10, 50, 60)), (1, 2, (100, 100, 120, 200)), (2, 1, (5, 5, 20, 20))]
    gt = _ds([1, 2], anns)
    dets = [_det(a[0], a[1], a[2], 0.9) for a in anns]
    s = evaluate_boxes(gt, dets)
    for k in ("AP", "AP50", "AP75", "AP60", "AP70", "AP80", "AR1", "AR10", "AR100"):
        assert abs(s[k] - 1.0) < 1e-12, (k, s[k])
    assert abs(s["APs"] - 1.0) < 1e-12 and abs(s["APm"] - 1.0) < 1e-12 and abs(s["APl"] - 1.0) < 1e-12
    assert evaluate_boxes(gt, [])["AP"] == 0.0        # nothing detected: every precision sample is 0


def test_hand_worked_precision_recall_curve():
    """One category, two ground truths, three detections by descending score: TP (IoU 1), FP, TP (IoU 0.81).
    IoU 0.5..0.80 (7 thresholds): tp/fp = 1,0 | 1,1 | 2,1 -> recall .5, .5, 1, precision 1, .5, 2/3 -> monotone
    1, 2/3, 2/3; sampled at recall 0..0.5 (51 thresholds) = 1 and 0.51..1.0 (50) = 2/3 -> AP_t = (51 + 50*2/3)/101.
    IoU 0.85..0.95 (3 thresholds): the third detection is a false positive: recall .5; AP_t = 51/101 (precision 1 up to
    recall 0.5, 0 beyond)."""
    from pet.rcnn.datasets.cocoeval import evaluate_boxes
    gt = _ds([1], [(1, 1, (0, 0, 100, 100)), (1, 1, (200, 200, 100, 100))], cats=(1,))
    dets = [_det(1, 1, (0, 0, 100, 100), 0.9), _det(1, 1, (400, 0, 50, 50), 0.8), _det(1, 1, (200, 200, 100, 81), 0.7)]
    s = evaluate_boxes(gt, dets)
    hi, lo = (51 + 50 * 2.0 / 3) / 101, 51.0 / 101
    assert abs(s["AP50"] - hi) < 1e-12 and abs(s["AP75"] - hi) < 1e-12 and abs(s["AP80"] - hi) < 1e-12
    assert abs(s["AP"] - (7 * hi + 3 * lo) / 10) < 1e-12
    assert abs(s["AR100"] - (7 * 1.0 + 3 * 0.5) / 10) < 1e-12 and abs(s["AR1"] - 0.5) < 1e-12


def test_crowd_and_area_ranges_and_maxdets():
    from pet.rcnn.datasets.cocoeval import evaluate_boxes
    # a crowd region absorbs any number of detections without penalty and is not counted as a ground truth
    gt = _ds([1], [(1, 1, (0, 0, 100, 100)), (1, 1, (300, 300, 200, 150), 1)], cats=(1,))
    dets = [_det(1, 1, (0, 0, 100, 100), 0.9), _det(1, 1, (310, 310, 30, 30), 0.8), _det(1, 1, (350, 350, 40, 40), 0.7)]
    s = evaluate_boxes(gt, dets)
    assert abs(s["AP"] - 1.0) < 1e-12 and abs(s["AR100"] - 1.0) < 1e-12
    # area ranges: a 20x20 (small) and a 200x200 (large) object, only the large one found
    gt = _ds([1], [(1, 1, (0, 0, 20, 20)), (1, 1, (100, 100, 200, 200))], cats=(1,))
    s = evaluate_boxes(gt, [_det(1, 1, (100, 100, 200, 200), 0.9)])
    assert s["APs"] == 0.0 and abs(s["APl"] - 1.0) < 1e-12 and s["APm"] == -1.0
    # maxDets: 12 objects all found -> AR1 = 1/12, AR10 = 10/12
    boxes = [(30 * i, 0, 20, 40) for i in range(12)]
    gt = _ds([1], [(1, 1, b) for b in boxes], cats=(1,))
    s = evaluate_boxes(gt, [_det(1, 1, b, 0.99 - 0.01 * i) for i, b in enumerate(boxes)])
    assert abs(s["AR1"] - 1 / 12) < 1e-12 and abs(s["AR10"] - 10 / 12) < 1e-12 and abs(s["AR100"] - 1.0) < 1e-12


def test_evaluation_entry_point_scores_without_pycocotools(tmp_path):
    """pet.rcnn.datasets.evaluation.evaluation writes bbox.json and returns the summary (evaluation.py:56-109)."""
    import json
    from pet.rcnn.core import config
    import importlib
    E = importlib.import_module("pet.rcnn.datasets.evaluation")
    gt = _ds([7], [(7, 1, (10, 10, 50, 60))], cats=(1,))
    ann = tmp_path / "ann.json"
    ann.write_text(json.dumps(gt))

    class DS(object):
        ann_file = str(ann)
    config.reset_cfg()
    config.cfg.CKPT = str(tmp_path)
    try:
        res, recs = E.evaluation(DS(), [_det(7, 1, (10, 10, 50, 60), 0.5)])
    finally:
        config.reset_cfg()
    assert abs(res["bbox"]["AP"] - 1.0) < 1e-12 and (tmp_path / "test" / "bbox.json").exists()


def test_equals_the_references_vendored_cocoeval():
    """tests/golden/cocoeval_ref.* (make_golden.py cocoeval): the reference's own COCOeval (pet/rcnn/datasets/
    mycocoeval.py:62-423, bbox protocol) run on 14 images x 3 categories with crowds, all three area ranges, jittered
    true positives, false positives and one (image, category) cell beyond maxDets -- only the box IoU routine it takes
    from pycocotools was supplied by the harness.  Precision [T,R,K,A,M] and recall [T,K,A,M] arrays must be EQUAL and
    the summary numbers agree to 1e-12 (AP90, which the reference prints from a hard-coded threshold index, included)."""
    import json
    import os
    from pet.rcnn.datasets.cocoeval import COCOBoxEval
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(here, "cocoeval_ref.json")) as f:
        ref = json.load(f)
    arr = np.load(os.path.join(here, "cocoeval_ref.npz"))
    ev = COCOBoxEval(ref["gt"], ref["dt"]).evaluate().accumulate()
    assert ev.eval["precision"].shape == arr["precision"].shape
    assert np.array_equal(ev.eval["precision"], arr["precision"])
    assert np.array_equal(ev.eval["recall"], arr["recall"])
    s = ev.summarize()
    names = ["AP", "AP50", "AP75", "APs", "APm", "APl", "AR1", "AR10", "AR100", "ARs", "ARm", "ARl", "AP60", "AP70", "AP80",
             "AP90"]
    assert len(ref["stats"]) == 16
    for n, want in zip(names, ref["stats"]):
        assert abs(s[n] - want) < 1e-12, (n, s[n], want)
