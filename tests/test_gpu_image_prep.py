"""GPU: cpm_image_prep (resize + flip + channel order + normalise + pad) against the oracle and against the host
chain itself (PIL resize + the ToTensor / Normalize tensor arithmetic) -- bit-exact."""
import json
import random

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
MEAN, STD = [102.9801, 115.9465, 122.7717], [1.0, 1.0, 1.0]


def _host_chain(im, oh, ow, flip, mean, std, bgr, pad_hw):
    pil = Image.fromarray(im).resize((ow, oh), Image.BILINEAR)
    if flip:
        pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
    t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float().div(255)
    if bgr:
        t = t[[2, 1, 0]] * 255
    t = t.sub_(torch.tensor(mean)[:, None, None]).div_(torch.tensor(std)[:, None, None])
    out = torch.zeros((3,) + tuple(pad_hw))
    out[:, :oh, :ow] = t
    return out


@pytest.mark.parametrize("h,w,oh,ow", [(48, 64, 80, 133), (97, 131, 40, 55), (60, 80, 60, 107), (33, 47, 100, 47),
                                       (64, 64, 64, 64), (50, 50, 7, 9), (5, 7, 64, 96)])
@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("nhwc", [True, False])
def test_image_prep_vs_oracle_and_host_chain(oracle, h, w, oh, ow, flip, nhwc):
    import pet.lib.ops as ops
    rng = np.random.default_rng(h + 7 * w)
    im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    pad = (oh + 5, ow + 11)
    std = [1.0, 57.375, 58.395]
    lut = ops.value_table(MEAN, std, True).cuda()
    batch = torch.full((2, 3) + pad, 7.0, device="cuda")
    if nhwc:
        batch = batch.contiguous(memory_format=torch.channels_last)
    ops.image_prep(torch.from_numpy(im).cuda(), (oh, ow), flip, lut, True, batch[1])
    got = batch[1].cpu()
    want = oracle.image_prep(im, (oh, ow), flip, MEAN, std, True, pad)
    assert np.array_equal(got.numpy(), want)
    assert torch.equal(got, _host_chain(im, oh, ow, flip, MEAN, std, True, pad))
    assert bool((batch[0] == 7.0).all())                        # the other slot is untouched


def test_image_prep_rgb_no_scale_and_errors():
    import pet.lib.ops as ops
    im = np.random.default_rng(1).integers(0, 256, (32, 40, 3), dtype=np.uint8)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    lut = ops.value_table(mean, std, False).cuda()
    dst = torch.empty((3, 64, 64), device="cuda")
    ops.image_prep(torch.from_numpy(im).cuda(), (48, 60), False, lut, False, dst)
    assert torch.equal(dst.cpu(), _host_chain(im, 48, 60, False, mean, std, False, (64, 64)))
    with pytest.raises(RuntimeError):
        ops.image_prep(torch.from_numpy(im), (48, 60), False, lut, False, dst)          # CPU tensor
    with pytest.raises(RuntimeError):
        ops.image_prep(torch.from_numpy(im).cuda(), (80, 60), False, lut, False, dst)   # slot too small
    with pytest.raises(RuntimeError):
        ops.image_prep(torch.from_numpy(im).cuda().float(), (48, 60), False, lut, False, dst)


def test_image_prep_full_size(oracle):
    """COCO-sized images into the BASELINE batch shape (800 x 1333 padded to 800 x 1344): one landscape up-scaled, one
    large image down-scaled (long taps), checked against PIL + the tensor arithmetic."""
    import pet.lib.ops as ops
    rng = np.random.default_rng(2)
    lut = ops.value_table(MEAN, STD, True).cuda()
    batch = torch.empty((2, 3, 800, 1344), device="cuda").contiguous(memory_format=torch.channels_last)
    cases = [(480, 640, 800, 1066, False), (2400, 3999, 800, 1333, True)]
    for i, (h, w, oh, ow, flip) in enumerate(cases):
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ops.image_prep(torch.from_numpy(im).cuda(), (oh, ow), flip, lut, True, batch[i])
        assert torch.equal(batch[i].cpu(), _host_chain(im, oh, ow, flip, MEAN, STD, True, (800, 1344)))


def test_loader_to_device_equals_host_chain(tmp_path):
    """COCODataset -> transforms -> BatchCollator -> images.to('cuda') through a real DataLoader with workers."""
    from pet.rcnn.core import config
    from pet.rcnn.datasets import build_transforms
    from pet.utils.data.collate_batch import BatchCollator
    from pet.utils.data.datasets import COCODataset
    from pet.utils.data.structures.image_list import ImageList
    rng = np.random.default_rng(5)
    images, anns, raw = [], [], {}
    for i, (h, w) in enumerate([(48, 64), (64, 48), (50, 80), (40, 44)]):
        px = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(px).save(str(tmp_path / ("im%d.png" % i)))
        raw[i] = px
        images.append({"id": i, "file_name": "im%d.png" % i, "height": h, "width": w})
        anns.append({"id": i + 1, "image_id": i, "bbox": [2.0, 3.0, 20.0, 25.0], "category_id": 1, "iscrowd": 0,
                     "area": 500})
    (tmp_path / "ann.json").write_text(json.dumps({"images": images, "annotations": anns,
                                                   "categories": [{"id": 1, "name": "a"}]}))
    config.reset_cfg()
    config.merge_cfg_from_list(["TRAIN.SCALES", (80, 96), "TRAIN.MAX_SIZE", 120])
    try:
        ds = COCODataset(str(tmp_path / "ann.json"), str(tmp_path), True, ("bbox",), build_transforms(True))
        loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=2,
                                             collate_fn=BatchCollator(32))
        seen = 0
        for deferred, targets, ids in loader:
            il = deferred.to("cuda")
            assert isinstance(il, ImageList) and il.tensors.is_contiguous(memory_format=torch.channels_last)
            assert il.tensors.shape[2] % 32 == 0 and il.tensors.shape[3] % 32 == 0
            for j, idx in enumerate(ids):
                im = deferred.images[j]
                oh, ow = im.out_hw
                assert tuple(il.image_sizes[j]) == (oh, ow) and targets[j].size == (ow, oh)
                want = _host_chain(raw[idx], oh, ow, im.flip, MEAN, STD, True, il.tensors.shape[2:])
                assert torch.equal(il.tensors[j].cpu(), want)
                seen += 1
        assert seen == 4
    finally:
        config.reset_cfg()
