"""Catalog lookups used by build_dataset (pet/rcnn/datasets/dataset_catalog.py:87-104)."""
from pet.utils.data.dataset_catalog import COMMON_DATASETS, _ANN_FN, _IM_DIR

_DATASETS = dict(COMMON_DATASETS)


def register(name, image_directory, annotation_file):
    """Add a COCO-format dataset at run time (tests, private data)."""
    _DATASETS[name] = {_IM_DIR: image_directory, _ANN_FN: annotation_file}


def datasets():
    return _DATASETS.keys()


def contains(name):
    return name in _DATASETS


def get_im_dir(name):
    return _DATASETS[name][_IM_DIR]


def get_ann_fn(name):
    return _DATASETS[name][_ANN_FN]
