from .coco import COCODataset, COCOIndex
from .concat_dataset import ConcatDataset
