"""COCO-format detection dataset (box annotations) for the CPM R-CNN path.

Counterpart of pet/utils/data/datasets/coco.py:46-117, which sits on torchvision's CocoDetection + pycocotools
(neither is a dependency here): `COCOIndex` is the small part of the pycocotools index the dataset uses.  Images are
decoded with PIL in the loader workers and stay uint8 (`DeferredImage`); every pixel transform runs on the MI355X
(pet/utils/data/collate_batch.py).  Mask / keypoint / parsing annotations are outside the CPM path.
"""
import json
import os

import numpy as np
import torch
from PIL import Image

from pet.utils.data.structures.bounding_box import BoxList
from pet.utils.data.transforms.transforms import DeferredImage


class COCOIndex(object):
    """imgs / anns / cats dictionaries and the image -> annotations map of a COCO json (annotation order kept)."""

    def __init__(self, annotation_file):
        with open(annotation_file, "r") as f:
            self.dataset = json.load(f)
        self.imgs = {im["id"]: im for im in self.dataset.get("images", [])}
        self.anns = {a["id"]: a for a in self.dataset.get("annotations", [])}
        self.cats = {c["id"]: c for c in self.dataset.get("categories", [])}
        self.imgToAnns = {}
        for a in self.dataset.get("annotations", []):
            self.imgToAnns.setdefault(a["image_id"], []).append(a)

    def getAnnIds(self, imgIds=(), iscrowd=None):
        imgIds = imgIds if isinstance(imgIds, (list, tuple)) else [imgIds]
        if len(imgIds) == 0:
            anns = self.dataset.get("annotations", [])
        else:
            anns = [a for i in imgIds for a in self.imgToAnns.get(i, [])]
        if iscrowd is not None:
            anns = [a for a in anns if a.get("iscrowd", 0) == iscrowd]
        return [a["id"] for a in anns]

    def loadAnns(self, ids):
        return [self.anns[i] for i in (ids if isinstance(ids, (list, tuple)) else [ids])]

    def getCatIds(self):
        return [c["id"] for c in self.dataset.get("categories", [])]

    def loadCats(self, ids):
        return [self.cats[i] for i in (ids if isinstance(ids, (list, tuple)) else [ids])]

    def loadImgs(self, ids):
        return [self.imgs[i] for i in (ids if isinstance(ids, (list, tuple)) else [ids])]


def has_valid_annotation(anno, filter_crowd=True):
    """An image trains only if it has a non-crowd box wider and taller than 1 px (coco.py:16-43, box part)."""
    if filter_crowd and len(anno) and "iscrowd" in anno[0]:
        anno = [o for o in anno if o["iscrowd"] == 0]
    if len(anno) == 0:
        return False
    return not all(any(v <= 1 for v in o["bbox"][2:]) for o in anno)


class COCODataset(torch.utils.data.Dataset):
    def __init__(self, ann_file, root, remove_images_without_annotations, ann_types=("bbox",), transforms=None):
        if tuple(ann_types) != ("bbox",):
            raise NotImplementedError("only box annotations are on the CPM R-CNN path, got %s" % (ann_types,))
        self.root = root
        self.ann_file = ann_file
        self.coco = COCOIndex(ann_file)
        self.ids = sorted(self.coco.imgs.keys())
        if remove_images_without_annotations:
            self.ids = [i for i in self.ids
                        if has_valid_annotation(self.coco.loadAnns(self.coco.getAnnIds(imgIds=i, iscrowd=None)))]
        cat_ids = self.coco.getCatIds()
        self.json_category_id_to_contiguous_id = {v: i + 1 for i, v in enumerate(cat_ids)}
        self.contiguous_category_id_to_json_id = {v: k for k, v in self.json_category_id_to_contiguous_id.items()}
        self.id_to_img_map = {k: v for k, v in enumerate(self.ids)}
        self.classes = ["__background__"] + [c["name"] for c in self.coco.loadCats(cat_ids)]
        self.ann_types = tuple(ann_types)
        self._transforms = transforms

    def __len__(self):
        return len(self.ids)

    def pull_image(self, index):
        """The decoded image as uint8 RGB [H,W,3] (the reference returns BGR via cv2, coco.py:119-130)."""
        info = self.coco.imgs[self.id_to_img_map[index]]
        with Image.open(os.path.join(self.root, info["file_name"])) as im:
            return np.asarray(im.convert("RGB"))

    def __getitem__(self, idx):
        img_id = self.ids[idx]
        anno = self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id))
        img = DeferredImage(self.pull_image(idx))
        if len(anno) and "iscrowd" in anno[0]:
            anno = [o for o in anno if o["iscrowd"] == 0]
        boxes = torch.as_tensor([o["bbox"] for o in anno], dtype=torch.float32).reshape(-1, 4)
        target = BoxList(boxes, img.size, mode="xywh").convert("xyxy")
        labels = torch.tensor([self.json_category_id_to_contiguous_id[o["category_id"]] for o in anno],
                              dtype=torch.int64)
        target.add_field("labels", labels)
        target = target.clip_to_image(remove_empty=True)
        if self._transforms is not None:
            img, target = self._transforms(img, target)
        return img, target, idx

    def get_img_info(self, index):
        return self.coco.imgs[self.id_to_img_map[index]]
