"""CPU: host logic of the input pipeline and its oracle.

* the numpy restatement of Pillow's bilinear resample is pinned bit-for-bit against PIL.Image.resize itself;
* the product's tap tables / value table (what cpm_image_prep consumes) equal the oracle's;
* Resize.get_size and the samplers reproduce sequences dumped from the reference's classes
  (tests/golden/data_pipeline.json, written by tests/golden/make_golden.py data);
* COCODataset + transforms + BatchCollator on a synthetic COCO-format dataset: targets, geometry, deferral.
"""
import json
import os
import random

import numpy as np
import pytest
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "data_pipeline.json")) as f:
        return json.load(f)


SIZES = [(48, 64, 80, 133), (97, 131, 40, 55), (60, 80, 60, 107), (33, 47, 100, 47), (240, 320, 400, 533),
         (250, 187, 667, 500), (50, 50, 7, 9), (5, 7, 64, 96)]


@pytest.mark.parametrize("h,w,oh,ow", SIZES)
def test_oracle_resize_is_pillow(oracle, h, w, oh, ow):
    rng = np.random.default_rng(h * 1000 + w)
    im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(im).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(oracle.pil_resize_bilinear(im, oh, ow), want)


@pytest.mark.parametrize("n_in,n_out", [(64, 133), (131, 55), (480, 800), (640, 1066), (1000, 1333), (2000, 800),
                                        (7, 96), (50, 7), (333, 333)])
def test_tap_tables_equal_oracle(oracle, n_in, n_out):
    from pet.lib.ops.image_prep import resample_tables
    b, k, ks = resample_tables(n_in, n_out)
    ob, ok = oracle.pil_bilinear_coeffs(n_in, n_out)
    assert ks == ok.shape[1] and b.dtype == np.int32 and k.dtype == np.int32
    assert np.array_equal(b, ob) and np.array_equal(k, ok)
    assert int(b[:, 0].min()) >= 0 and int((b[:, 0] + b[:, 1]).max()) <= n_in and int(b[:, 1].max()) <= ks


def test_value_table_is_totensor_normalize():
    from pet.lib.ops.image_prep import value_table
    mean, std = [102.9801, 115.9465, 122.7717], [1.0, 57.375, 58.395]
    lut = value_table(mean, std, True)
    v = torch.arange(256, dtype=torch.uint8).view(1, 1, 256).repeat(3, 1, 1)
    t = v.float().div(255)                                      # ToTensor
    t = t[[2, 1, 0]] * 255                                      # Normalize.to_bgr255
    t = t.sub_(torch.tensor(mean)[:, None, None]).div_(torch.tensor(std)[:, None, None])
    assert torch.equal(lut, t[:, 0, :])
    lut2 = value_table([0.485, 0.456, 0.406], [0.229, 0.224, 0.225], False)
    t2 = v.float().div(255).sub_(torch.tensor([0.485, 0.456, 0.406])[:, None, None]).div_(
        torch.tensor([0.229, 0.224, 0.225])[:, None, None])
    assert torch.equal(lut2, t2[:, 0, :])


def test_resize_get_size_matches_reference(golden):
    from pet.utils.data.transforms import Resize
    for w, h, mins, mx, seed, want in golden["get_size"]:
        random.seed(seed)
        assert list(Resize(tuple(mins), mx).get_size((w, h))) == want


def test_samplers_match_reference(golden):
    from pet.utils.data.samplers import DistributedSampler, GroupedBatchSampler, IterationBasedBatchSampler
    for n, world, r, epoch, shuffle, want in golden["distributed"]:
        s = DistributedSampler(list(range(n)), num_replicas=world, rank=r, shuffle=shuffle)
        s.set_epoch(epoch)
        assert list(s) == want and len(s) == len(want)
    for n, world, r, epoch, bs, drop, gids, want, length in golden["grouped"]:
        s = DistributedSampler(list(range(n)), num_replicas=world, rank=r, shuffle=True)
        s.set_epoch(epoch)
        b = GroupedBatchSampler(s, gids, bs, drop_uneven=drop)
        assert len(b) == length
        assert [list(x) for x in b] == want
        for batch in b:                                         # the property the sampler exists for
            assert len({gids[i] for i in batch}) == 1
    for n, bs, iters, start, gids, want in golden["iteration"]:
        s = DistributedSampler(list(range(n)), num_replicas=1, rank=0, shuffle=True)
        it = IterationBasedBatchSampler(GroupedBatchSampler(s, gids, bs), iters, start)
        assert [list(x) for x in it] == want and len(it) == iters


def test_range_sampler_and_empty_group():
    from pet.utils.data.samplers import GroupedBatchSampler, RangeSampler
    assert list(RangeSampler(3, 7)) == [3, 4, 5, 6] and len(RangeSampler(3, 7)) == 4
    # a group with no sampled member (the reference raises IndexError there) simply yields no batch
    b = GroupedBatchSampler(RangeSampler(0, 3), [0, 0, 0, 1, 1], 2)
    assert [list(x) for x in b] == [[0, 1], [2]]


@pytest.fixture()
def tiny_coco(tmp_path):
    rng = np.random.default_rng(5)
    images, anns = [], []
    sizes = [(48, 64), (64, 48), (50, 80), (40, 40)]
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(str(tmp_path / ("im%d.png" % i)))
        images.append({"id": 10 + i, "file_name": "im%d.png" % i, "height": h, "width": w})
    anns.append({"id": 1, "image_id": 10, "bbox": [4.0, 6.0, 20.0, 30.0], "category_id": 3, "iscrowd": 0, "area": 600})
    anns.append({"id": 2, "image_id": 10, "bbox": [1, 1, 10, 10], "category_id": 7, "iscrowd": 1, "area": 100})
    anns.append({"id": 3, "image_id": 11, "bbox": [10.0, 10.0, 60.0, 20.0], "category_id": 7, "iscrowd": 0, "area": 1200})
    anns.append({"id": 4, "image_id": 12, "bbox": [5.0, 5.0, 1.0, 20.0], "category_id": 3, "iscrowd": 0, "area": 20})
    ann_file = tmp_path / "ann.json"
    ann_file.write_text(json.dumps({"images": images, "annotations": anns,
                                    "categories": [{"id": 3, "name": "a"}, {"id": 7, "name": "b"}]}))
    return str(tmp_path), str(ann_file)


def test_coco_dataset_and_collate(tiny_coco, oracle):
    from pet.rcnn.core import config
    from pet.rcnn.datasets import build_transforms
    from pet.utils.data.collate_batch import BatchCollator, DeferredBatch
    from pet.utils.data.datasets import COCODataset
    root, ann = tiny_coco
    config.reset_cfg()
    config.merge_cfg_from_list(["TRAIN.SCALES", (80,), "TRAIN.MAX_SIZE", 120])
    try:
        ds_all = COCODataset(ann, root, False, ("bbox",), None)
        assert len(ds_all) == 4 and ds_all.classes == ["__background__", "a", "b"]
        ds = COCODataset(ann, root, True, ("bbox",), build_transforms(True))
        assert ds.ids == [10, 11]                       # 12: only a 1-px-wide box; 13: no annotation
        assert ds.get_img_info(1)["file_name"] == "im1.png"
        random.seed(3)
        samples = [ds[0], ds[1]]
        for (img, tgt, idx), (h, w) in zip(samples, [(48, 64), (64, 48)]):
            assert img.pixels.shape == (h, w, 3) and img.pixels.dtype == np.uint8
            assert tgt.size == img.size and tgt.mode == "xyxy" and len(tgt) == 1
        img0, tgt0, _ = samples[0]
        assert img0.out_hw == (80, 106)                 # shorter side 48 -> 80, 64 * 80 / 48 = 106.67 -> 106
        box = torch.tensor([[4.0, 6.0, 23.0, 35.0]]) * torch.tensor([106 / 64., 80 / 48., 106 / 64., 80 / 48.])
        if img0.flip:
            box = torch.stack([106 - box[:, 2] - 1, box[:, 1], 106 - box[:, 0] - 1, box[:, 3]], 1)
        assert torch.allclose(tgt0.bbox, box, atol=1e-4) and tgt0.get_field("labels").tolist() == [1]
        images, targets, ids = BatchCollator(32)(samples)
        assert isinstance(images, DeferredBatch) and images.batch_hw == (128, 128) and ids == (0, 1)
        assert [tuple(s) for s in images.image_sizes] == [(80, 106), (106, 80)]
        with pytest.raises(RuntimeError):
            images.to("cpu")
    finally:
        config.reset_cfg()


def test_transform_chain_order_is_enforced():
    from pet.utils.data import transforms as T
    im = T.DeferredImage(np.zeros((8, 8, 3), np.uint8))
    with pytest.raises(RuntimeError):
        T.Normalize([0, 0, 0], [1, 1, 1])(im, None)
    with pytest.raises(NotImplementedError):
        T.ColorJitter(brightness=0.2)
    with pytest.raises(TypeError):
        T.DeferredImage(np.zeros((8, 8), np.uint8))
