"""ImageList / to_image_list (counterpart of pet/utils/data/structures/image_list.py:4-69): a zero-padded
batch tensor plus the unpadded (height, width) of each image."""
import math

import torch


class ImageList(object):
    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def to_image_list(tensors, size_divisible=0):
    if isinstance(tensors, torch.Tensor) and size_divisible > 0:
        tensors = [tensors]
    if isinstance(tensors, ImageList):
        return tensors
    if isinstance(tensors, torch.Tensor):
        if tensors.dim() == 3:
            tensors = tensors[None]
        assert tensors.dim() == 4
        return ImageList(tensors, [t.shape[-2:] for t in tensors])
    if isinstance(tensors, (tuple, list)):
        c, h, w = (max(s) for s in zip(*[img.shape for img in tensors]))
        if size_divisible > 0:
            h = int(math.ceil(h / size_divisible) * size_divisible)
            w = int(math.ceil(w / size_divisible) * size_divisible)
        batch = tensors[0].new_zeros((len(tensors), c, h, w))
        for img, slot in zip(tensors, batch):
            slot[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        return ImageList(batch, [im.shape[-2:] for im in tensors])
    raise TypeError("Unsupported type for to_image_list: {}".format(type(tensors)))
