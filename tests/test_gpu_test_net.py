"""GPU: the test-time path -- device resize of the image blob (cv2.resize INTER_LINEAR semantics), im_detect_bbox with
flip / scale augmentation, and tools/rcnn/test_net.py end to end on a synthetic COCO-format dataset."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import ROOT
from test_host_logic import CASCADE_OPTS, CPM_OPTS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,scale", [(48, 64, 1.6667), (97, 131, 0.41), (60, 80, 1.0), (33, 47, 2.5), (480, 640, 1.25)])
@pytest.mark.parametrize("flip", [False, True])
def test_resize_linear_vs_oracle(oracle, h, w, scale, flip):
    import pet.lib.ops as ops
    rng = np.random.default_rng(h + w)
    im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = ops.resize_linear(torch.from_numpy(im).cuda(), scale, flip=flip, swap_rb=True).cpu().numpy()
    want = oracle.cv_resize_linear(im, scale, flip)[:, :, ::-1].transpose(2, 0, 1)          # RGB -> BGR planes
    assert got.shape == want.shape == (3, int(np.rint(h * scale)), int(np.rint(w * scale)))
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
    if scale == 1.0:
        src = im[:, ::-1] if flip else im
        assert np.array_equal(got, src[:, :, ::-1].transpose(2, 0, 1).astype(np.float32))


def _make_dataset(root, n=3):
    rng = np.random.default_rng(13)
    images, anns = [], []
    for i in range(n):
        h, w = (120, 160) if i % 2 == 0 else (160, 120)
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, "im%d.png" % i))
        images.append({"id": 100 + i, "file_name": "im%d.png" % i, "height": h, "width": w})
        anns.append({"id": i + 1, "image_id": 100 + i, "bbox": [10.0, 12.0, 60.0, 50.0], "category_id": 5 + i,
                     "iscrowd": 0, "area": 3000.0})
    with open(os.path.join(root, "ann.json"), "w") as f:
        json.dump({"images": images, "annotations": anns,
                   "categories": [{"id": c, "name": "c%d" % c} for c in range(1, 81)]}, f)


def _run_test_net(tmp_path, opts, name):
    from pet.rcnn.core import config
    from pet.rcnn.datasets import dataset_catalog
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.checkpointer import CheckPointer
    spec = importlib.util.spec_from_file_location("test_net", os.path.join(ROOT, "tools", "rcnn", "test_net.py"))
    test_net = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(test_net)
    data = tmp_path / "data"
    data.mkdir()
    _make_dataset(str(data))
    dataset_catalog.register(name, str(data), str(data / "ann.json"))
    ckpt = str(tmp_path / "ckpt")
    config.reset_cfg()
    config.merge_cfg_from_list(opts)
    torch.manual_seed(0)
    CheckPointer(ckpt, auto_resume=False).save(Generalized_RCNN(is_train=True), copy_latest=False)   # BN form, as trained
    config.reset_cfg()
    argv = [str(o) for o in opts] + ["TEST.DATASETS", str((name,)), "TEST.SCALE", "160", "TEST.MAX_SIZE", "256",
                                    "CKPT", ckpt]
    res = test_net.main(argv)
    with open(os.path.join(ckpt, "test", "bbox.json")) as f:
        recs = json.load(f)
    assert os.path.exists(os.path.join(ckpt, "test", "detections.pkl"))
    return res, recs


def test_test_net_cpm_end_to_end(tmp_path):
    from pet.rcnn.core import config
    try:
        res, recs = _run_test_net(tmp_path, list(CPM_OPTS) + ["GRID_RCNN.SCORE_THRESH", 0.0125], "synthetic_val_cpm")
        assert isinstance(recs, list)
        sizes = {100: (160, 120), 101: (120, 160), 102: (160, 120)}
        for r in recs:
            assert r["image_id"] in sizes and 1 <= r["category_id"] <= 80 and np.isfinite(r["score"])
            assert len(r["bbox"]) == 4 and all(np.isfinite(v) for v in r["bbox"])
    finally:
        config.reset_cfg()


def test_test_net_cascade_with_tta_soft_nms_voting(tmp_path):
    """Offset-regression cascade through filter_results with flip + scale augmentation, soft-NMS and box voting."""
    from pet.rcnn.core import config
    try:
        opts = list(CASCADE_OPTS) + ["FAST_RCNN.SCORE_THRESH", 0.013, "FAST_RCNN.DETECTIONS_PER_IMG", 20,
                                      "TEST.BBOX_AUG.ENABLED", True, "TEST.BBOX_AUG.H_FLIP", True,
                                      "TEST.BBOX_AUG.SCALES", (128,), "TEST.BBOX_AUG.MAX_SIZE", 300,
                                      "TEST.SOFT_NMS.ENABLED", True, "TEST.BBOX_VOTE.ENABLED", True]
        res, recs = _run_test_net(tmp_path, opts, "synthetic_val_cascade")
        per_image = {}
        for r in recs:
            per_image[r["image_id"]] = per_image.get(r["image_id"], 0) + 1
            assert 1 <= r["category_id"] <= 80 and np.isfinite(r["score"]) and all(np.isfinite(v) for v in r["bbox"])
        assert len(recs) > 0 and all(v <= 20 + 5 for v in per_image.values())      # kthvalue keeps ties
    finally:
        config.reset_cfg()
