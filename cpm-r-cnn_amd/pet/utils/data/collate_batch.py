"""BatchCollator (pet/utils/data/collate_batch.py:4-20) for deferred images.

The reference's collator pads fp32 tensors into the batch on the host; this one only records the padded batch
geometry.  `DeferredBatch.to(device)` -- the call the training loop already makes on the collated images
(tools/rcnn/train_net.py:66 `images.to(device)`) -- uploads the decoded uint8 pixels of the whole batch in ONE copy from
a pinned staging buffer and produces every padded fp32 image slot on the MI355X (cpm_image_prep), returning the
ImageList the model consumes.
"""
import math

import numpy as np
import torch

from pet.utils.data.structures.image_list import ImageList, to_image_list
from pet.utils.data.transforms.transforms import DeferredImage


class _Staging(object):
    """Three rotating pinned buffers; a buffer is reused only after the copy that read it has completed."""

    def __init__(self):
        self.bufs = [None, None, None]
        self.events = [None, None, None]
        self.i = 0

    def get(self, nbytes):
        self.i = (self.i + 1) % 3
        if self.events[self.i] is not None:
            self.events[self.i].synchronize()
        b = self.bufs[self.i]
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8).pin_memory()
            self.bufs[self.i] = b
        return b, self.i


_staging = _Staging()
_luts = {}


class DeferredBatch(object):
    """A collated batch whose pixels are still uint8 on the host."""

    def __init__(self, images, size_divisible=0):
        self.images = list(images)
        for im in self.images:
            if not (im.as_tensor and im.norm is not None):
                raise RuntimeError("the transform chain must end with ToTensor + Normalize (build_transforms)")
        h = max(im.out_hw[0] for im in self.images)
        w = max(im.out_hw[1] for im in self.images)
        if size_divisible > 0:                                 # image_list.py:47-53
            h = int(math.ceil(h / size_divisible) * size_divisible)
            w = int(math.ceil(w / size_divisible) * size_divisible)
        self.batch_hw = (h, w)
        self.image_sizes = [torch.Size(im.out_hw) for im in self.images]

    def __len__(self):
        return len(self.images)

    def to(self, device, memory_format=torch.channels_last, **_):
        from pet.lib.ops.image_prep import image_prep, value_table
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("DeferredBatch.to: images are prepared on the MI355X only (no CPU fallback)")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        sizes = [im.pixels.size for im in self.images]
        offs = np.concatenate([[0], np.cumsum([(s + 255) // 256 * 256 for s in sizes])]).astype(np.int64)
        host, slot = _staging.get(int(offs[-1]))
        hv = host.numpy()
        for im, o, s in zip(self.images, offs[:-1], sizes):
            hv[o:o + s] = im.pixels.reshape(-1)
        dev = host[: int(offs[-1])].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        _staging.events[slot] = ev
        h, w = self.batch_hw
        batch = torch.empty((len(self.images), 3, h, w), dtype=torch.float32, device=device,
                            memory_format=memory_format)
        for i, (im, o, s) in enumerate(zip(self.images, offs[:-1], sizes)):
            key = (im.norm, device.index)
            lut = _luts.get(key)
            if lut is None:
                lut = _luts[key] = value_table(*im.norm).to(device)
            src = dev[int(o): int(o) + s].view(im.pixels.shape)
            image_prep(src, im.out_hw, im.flip, lut, im.norm[2], batch[i])
        return ImageList(batch, self.image_sizes)


class BatchCollator(object):
    def __init__(self, size_divisible=0):
        self.size_divisible = size_divisible

    def __call__(self, batch):
        images, targets, img_ids = list(zip(*batch))[:3]
        if all(isinstance(im, DeferredImage) for im in images):
            images = DeferredBatch(images, self.size_divisible)
        else:
            images = to_image_list(images, self.size_divisible)
        return images, targets, img_ids
