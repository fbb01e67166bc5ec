"""GPU: the training driver end to end on a synthetic COCO-format dataset -- loader workers (uint8 decode), device
image preparation, CPM R-CNN forward/backward, flat SGD, snapshots, auto-resume (tools/rcnn/train_net.py)."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import ROOT
from test_host_logic import CPM_OPTS

pytestmark = pytest.mark.gpu


def _make_dataset(root, n=8):
    rng = np.random.default_rng(11)
    images, anns = [], []
    for i in range(n):
        h, w = (200, 260) if i % 3 else (260, 200)
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, "im%d.png" % i))
        images.append({"id": i + 1, "file_name": "im%d.png" % i, "height": h, "width": w})
        for j in range(3):
            bw, bh = rng.uniform(40, 120), rng.uniform(40, 120)
            x, y = rng.uniform(0, w - bw - 1), rng.uniform(0, h - bh - 1)
            anns.append({"id": len(anns) + 1, "image_id": i + 1, "bbox": [float(x), float(y), float(bw), float(bh)],
                         "category_id": int(rng.integers(1, 81)), "iscrowd": 0, "area": float(bw * bh)})
    with open(os.path.join(root, "ann.json"), "w") as f:
        json.dump({"images": images, "annotations": anns,
                   "categories": [{"id": c, "name": "c%d" % c} for c in range(1, 81)]}, f)


def test_train_net_runs_snapshots_and_resumes(tmp_path):
    from pet.rcnn.core import config
    from pet.rcnn.datasets import dataset_catalog
    spec = importlib.util.spec_from_file_location("train_net", os.path.join(ROOT, "tools", "rcnn", "train_net.py"))
    train_net = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(train_net)
    data = tmp_path / "data"
    data.mkdir()
    _make_dataset(str(data))
    dataset_catalog.register("synthetic_train", str(data), str(data / "ann.json"))
    ckpt = str(tmp_path / "ckpt")
    opts = list(CPM_OPTS) + ["TRAIN.DATASETS", ("synthetic_train",), "TRAIN.SCALES", (200,), "TRAIN.MAX_SIZE", 300,
                              "TRAIN.BATCH_SIZE", 2, "TRAIN.LOADER_THREADS", 2, "SOLVER.BASE_LR", 0.0005,
                              "SOLVER.WARM_UP_ITERS", 2, "SOLVER.SNAPSHOT_ITERS", 2, "DISPLAY_ITER", 1, "CKPT", ckpt]
    opts = [str(o) for o in opts]                       # as typed on the command line
    config.reset_cfg()
    try:
        torch.manual_seed(0)
        model = train_net.main(opts + ["SOLVER.MAX_ITER", "4"])
        assert sorted(os.listdir(ckpt)) == ["model_iter2.pth", "model_iter4.pth", "model_latest.pth"]
        blob = torch.load(os.path.join(ckpt, "model_latest.pth"), map_location="cpu", weights_only=False)
        assert blob["scheduler"]["iteration"] == 4 and len(blob["optimizer"]["state"]) == 196
        sd = model.state_dict()
        for k, v in blob["model"].items():
            assert torch.equal(v, sd[k].detach().cpu()), k
        assert tuple(blob["model"]["Grid_Cascade_RCNN.Head_cls.fc6.weight"].shape) == (1024, 12544)
        w4 = blob["model"]["Conv_Body.layer2.0.conv1.weight"].clone()
        # second run: picks model_latest.pth up, continues at iteration 4 and stops at 6
        config.reset_cfg()
        model2 = train_net.main(opts + ["SOLVER.MAX_ITER", "6"])
        assert "model_iter6.pth" in os.listdir(ckpt)
        blob2 = torch.load(os.path.join(ckpt, "model_latest.pth"), map_location="cpu", weights_only=False)
        assert blob2["scheduler"]["iteration"] == 6
        assert not torch.equal(blob2["model"]["Conv_Body.layer2.0.conv1.weight"], w4)        # training went on
        for v in blob2["model"].values():
            assert torch.isfinite(v).all()
        del model, model2
    finally:
        config.reset_cfg()
