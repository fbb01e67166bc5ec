"""Detections -> COCO result records and their evaluation (counterpart of pet/rcnn/datasets/evaluation.py:21-153, box
part).  `post_processing` maps the boxes of a batch back to the original image sizes and emits COCO json records;
`evaluation` writes `<CKPT>/test/bbox.json` and scores it with pet.rcnn.datasets.cocoeval (the published COCOeval
algorithm restated in numpy: the reference evaluates with its own copy of that class, mycocoeval.py, over pycocotools'
compiled IoU code, and pycocotools is absent from the reference tree and from this image)."""
import json
import logging
import os

import numpy as np
import torch

from pet.rcnn.core.config import cfg

_log = logging.getLogger("pet.evaluation")


def prepare_box_results(results, image_ids, dataset):
    box_results, ims_dets, ims_labels = [], [], []
    if cfg.MODEL.RPN_ONLY:
        return results, None, None
    for result, image_id in zip(results, image_ids):
        original_id = dataset.id_to_img_map[image_id]
        if len(result) == 0:
            ims_dets.append(None)
            ims_labels.append(None)
            continue
        info = dataset.get_img_info(image_id)
        result = result.resize((info["width"], info["height"]))
        scores = result.get_field("scores")
        labels = result.get_field("labels").tolist()
        ims_dets.append(np.hstack((result.bbox.numpy(), scores.numpy()[:, np.newaxis])).astype(np.float32, copy=False))
        ims_labels.append(labels)
        boxes = result.convert("xywh").bbox.tolist()
        scores = scores.tolist()
        box_results.extend({"image_id": original_id, "category_id": dataset.contiguous_category_id_to_json_id[labels[k]],
                            "bbox": box, "score": scores[k]} for k, box in enumerate(boxes))
    return box_results, ims_dets, ims_labels


def post_processing(results, image_ids, dataset):
    results = [o.to(torch.device("cpu")) for o in results]
    box_results, ims_dets, ims_labels = prepare_box_results(results, image_ids, dataset)
    none = [None for _ in image_ids]
    return [box_results, [], [], [], [], []], [ims_dets, ims_labels, none, none, none, none]


def evaluation(dataset, all_boxes, *unused):
    out_dir = os.path.join(cfg.CKPT, "test")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "bbox.json")
    with open(path, "w") as f:
        json.dump(all_boxes, f)
    _log.info("Wrote %d detections to %s", len(all_boxes), path)
    from pet.rcnn.datasets.cocoeval import evaluate_boxes
    ann_file = getattr(dataset, "ann_file", None)
    with open(ann_file) as f:
        gt = json.load(f)
    stats = evaluate_boxes(gt, all_boxes)
    _log.info("bbox: %s", ", ".join("%s %.4f" % kv for kv in stats.items()))
    return {"bbox": dict(stats)}, {"bbox": all_boxes}
