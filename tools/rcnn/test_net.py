#!/usr/bin/env python3
"""Inference / evaluation driver with the reference's command line (tools/rcnn/test_net.py:10-55):

    python tools/rcnn/test_net.py --cfg cfgs/...yaml [--range START END] [KEY VALUE ...]

Loads TEST.WEIGHTS (or <CKPT>/model_latest.pth), runs the test-time path on TEST.DATASETS on one MI355X (images are
resized on the device), writes <CKPT>/test/{detections.pkl,bbox.json} and scores them when pycocotools is available."""
import argparse
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))

from pet.rcnn.core.config import assert_and_infer_cfg, cfg, merge_cfg_from_file, merge_cfg_from_list  # noqa: E402
from pet.rcnn.core.test_engine import run_inference  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser(description="CPM R-CNN testing on MI355X")
    p.add_argument("--cfg", dest="cfg_file", default=None, type=str)
    p.add_argument("--gpu_id", type=str, default="0", help="kept for command-line compatibility (one process per GPU)")
    p.add_argument("--range", help="start (inclusive) and end (exclusive) indices", type=int, nargs=2)
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    args = p.parse_args(argv)
    if args.cfg_file:
        merge_cfg_from_file(args.cfg_file)
    if args.opts:
        merge_cfg_from_list(args.opts)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    os.makedirs(os.path.join(cfg.CKPT or ".", "test"), exist_ok=True)
    assert_and_infer_cfg(make_immutable=False)
    return run_inference(ind_range=args.range)


if __name__ == "__main__":
    main()
