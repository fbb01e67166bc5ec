"""Inference over a dataset (counterpart of pet/rcnn/core/test_engine.py:21-214, box path): build the test model from
cfg, run im_detect_bbox image batch by image batch, turn the results into COCO records, evaluate.  One process per GPU;
a sub-range of the dataset can be given (the reference's multi-GPU mode spawns one such range per GPU)."""
import logging
import os
import pickle
import time

import torch

import pet.rcnn.core.test as rcnn_test
from pet.rcnn.core.config import cfg
from pet.rcnn.datasets import build_dataset, evaluation, post_processing
from pet.rcnn.modeling.model_builder import Generalized_RCNN
from pet.utils.checkpointer import get_weights, load_weights
from pet.utils.net import convert_bn2affine_model

_log = logging.getLogger("pet.test_engine")


def initialize_model_from_cfg():
    """Test-mode model with the trained weights (test_engine.py:200-214)."""
    model = Generalized_RCNN(is_train=False)
    cfg.TEST.WEIGHTS = get_weights(cfg.CKPT, cfg.TEST.WEIGHTS)
    load_weights(model, cfg.TEST.WEIGHTS)
    if cfg.MODEL.BATCH_NORM == "freeze":
        model = convert_bn2affine_model(model)
    model.eval()
    return model.to(torch.device(cfg.DEVICE)).to(memory_format=torch.channels_last)


def test(model, dataset, start_ind, end_ind):
    all_boxes = []
    step = cfg.TEST.IMS_PER_GPU
    t0 = time.time()
    for i in range(start_ind, end_ind, step):
        ids = list(range(i, min(i + step, end_ind)))
        ims = [dataset.pull_image(j) for j in ids]
        result, _ = rcnn_test.im_detect_bbox(model, ims)
        (box_results, *_), _ = post_processing(result, ids, dataset)
        all_boxes += box_results
        done = ids[-1] + 1 - start_ind
        if done % 100 < step:
            _log.info("%d / %d images, %.2f img/s", done, end_ind - start_ind, done / (time.time() - t0))
    return all_boxes


def test_net(ind_range=None, model=None):
    dataset = build_dataset(cfg.TEST.DATASETS, is_train=False)
    start_ind, end_ind = ind_range if ind_range is not None else (0, len(dataset))
    model = model if model is not None else initialize_model_from_cfg()
    all_boxes = test(model, dataset, start_ind, end_ind)
    name = "detection_range_%s_%s.pkl" % tuple(ind_range) if ind_range is not None else "detections.pkl"
    os.makedirs(os.path.join(cfg.CKPT, "test"), exist_ok=True)
    with open(os.path.join(cfg.CKPT, "test", name), "wb") as f:
        pickle.dump(dict(all_boxes=all_boxes), f)
    return dataset, all_boxes


def run_inference(ind_range=None, model=None):
    dataset, all_boxes = test_net(ind_range, model)
    if ind_range is not None:
        return None, {"bbox": all_boxes}          # a partial range is only stored (test_engine.py:117-133)
    return evaluation(dataset, all_boxes)
