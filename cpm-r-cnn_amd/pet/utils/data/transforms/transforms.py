"""Image/target transforms with the reference's class surface (pet/utils/data/transforms/transforms.py:11-115),
re-designed so that no pixel is touched on the host.

The reference resizes, flips, converts and normalises every PIL image in the loader workers and ships the fp32
result.  Here the transforms run on a `DeferredImage` -- the decoded uint8 pixels plus the *parameters* the chain
decides (output size, flip, value table) -- while the targets (BoxList) are transformed eagerly exactly as in the
reference.  The pixels are produced on the MI355X by `DeferredBatch.to(device)` (collate_batch.py) through
cpm_image_prep, bit-identical to the host chain.
"""
import random

import numpy as np


class DeferredImage(object):
    """Decoded RGB pixels (uint8 [H,W,3]) + the pending geometry / value transforms."""

    def __init__(self, pixels):
        pixels = np.asarray(pixels)
        if pixels.dtype != np.uint8 or pixels.ndim != 3 or pixels.shape[2] != 3:
            raise TypeError("DeferredImage needs RGB uint8 [H,W,3] pixels (PIL.Image.convert('RGB'))")
        self.pixels = np.ascontiguousarray(pixels)
        self.out_hw = (int(pixels.shape[0]), int(pixels.shape[1]))
        self.flip = False
        self.as_tensor = False
        self.norm = None                       # (mean, std, to_bgr255)

    @property
    def size(self):
        """(width, height) of the image as the chain has shaped it so far -- PIL.Image.size."""
        return (self.out_hw[1], self.out_hw[0])

    @property
    def shape(self):
        """(3, H, W): what the tensor will be (to_image_list reads .shape)."""
        return (3, self.out_hw[0], self.out_hw[1])


def _deferred(image):
    if isinstance(image, DeferredImage):
        return image
    return DeferredImage(np.asarray(image.convert("RGB") if hasattr(image, "convert") else image))


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target):
        image = _deferred(image)
        for t in self.transforms:
            image, target = t(image, target)
        return image, target

    def __repr__(self):
        return self.__class__.__name__ + "(" + "".join("\n    {0}".format(t) for t in self.transforms) + "\n)"


class Resize(object):
    def __init__(self, min_size, max_size):
        if not isinstance(min_size, (list, tuple)):
            min_size = (min_size,)
        self.min_size = min_size
        self.max_size = max_size

    def get_size(self, image_size):
        """(oh, ow) for a (w, h) image: shorter side -> a random choice of min_size unless the longer side would
        exceed max_size (transforms.py:38-58)."""
        w, h = image_size
        size = random.choice(self.min_size)
        max_size = self.max_size
        if max_size is not None:
            lo, hi = float(min((w, h))), float(max((w, h)))
            if hi / lo * size > max_size:
                size = int(round(max_size * lo / hi))
        if (w <= h and w == size) or (h <= w and h == size):
            return (h, w)
        if w < h:
            return (int(size * h / w), size)
        return (size, int(size * w / h))

    def __call__(self, image, target):
        image = _deferred(image)
        if image.flip or image.out_hw != tuple(image.pixels.shape[:2]):
            raise RuntimeError("Resize must be the first geometric transform of the chain (as in build_transforms)")
        image.out_hw = tuple(int(v) for v in self.get_size(image.size))
        if target is not None:
            target = target.resize(image.size)
        return image, target


class RandomHorizontalFlip(object):
    def __init__(self, prob=0.5, left_right=()):
        self.prob = prob
        self.left_right = left_right

    def __call__(self, image, target):
        if random.random() < self.prob:
            image = _deferred(image)
            image.flip = not image.flip
            if target is not None:
                target = target.transpose(0)
        return image, target


class ColorJitter(object):
    """The CPM configs leave brightness/contrast/saturation/hue at 0 (identity); non-zero jitter is not part of the
    device chain."""

    def __init__(self, brightness=None, contrast=None, saturation=None, hue=None):
        if any(v not in (None, 0, 0.0) for v in (brightness, contrast, saturation, hue)):
            raise NotImplementedError("ColorJitter with non-zero parameters is outside the CPM R-CNN path")

    def __call__(self, image, target):
        return image, target


class ToTensor(object):
    def __call__(self, image, target):
        image = _deferred(image)
        image.as_tensor = True
        return image, target


class Normalize(object):
    def __init__(self, mean, std, to_bgr255=True):
        self.mean = np.asarray(mean, dtype=np.float64).reshape(-1)
        self.std = np.asarray(std, dtype=np.float64).reshape(-1)
        self.to_bgr255 = to_bgr255

    def __call__(self, image, target):
        image = _deferred(image)
        if not image.as_tensor:
            raise RuntimeError("Normalize comes after ToTensor")
        image.norm = (tuple(self.mean.tolist()), tuple(self.std.tolist()), bool(self.to_bgr255))
        return image, target
