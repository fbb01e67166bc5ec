"""Factories selected by cfg strings (counterpart of pet/rcnn/modeling/registry.py:4-27, hot-path ones)."""
from pet.utils.registry import Registry

BACKBONES = Registry()
FPN_BODY = Registry()
ROI_CLS_HEADS = Registry()
ROI_CLS_OUTPUTS = Registry()
ROI_GRID_HEADS = Registry()
ROI_GRID_OUTPUTS = Registry()
ROI_CASCADE_HEADS = Registry()
ROI_CASCADE_OUTPUTS = Registry()
