"""Dataset name -> (image directory, annotation file), the COCO entries of pet/utils/data/dataset_catalog.py:4-60.
The data root is <repo>/data unless CPM_DATA_DIR is set."""
import os
import os.path as osp

ROOT_DIR = osp.abspath(osp.join(osp.dirname(__file__), "..", "..", ".."))
_DATA_DIR = os.environ.get("CPM_DATA_DIR", osp.abspath(osp.join(ROOT_DIR, "data")))
_IM_DIR = "image_directory"
_ANN_FN = "annotation_file"


def _coco(split, ann):
    return {_IM_DIR: _DATA_DIR + "/coco/images/" + split, _ANN_FN: _DATA_DIR + "/coco/annotations/" + ann}


COMMON_DATASETS = {
    "coco_2017_train": _coco("train2017", "instances_train2017.json"),
    "coco_2017_val": _coco("val2017", "instances_val2017.json"),
    "coco_2017_test": _coco("test2017", "image_info_test2017.json"),
    "coco_2017_test-dev": _coco("test2017", "image_info_test-dev2017.json"),
}
