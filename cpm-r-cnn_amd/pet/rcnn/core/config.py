"""Global cfg tree for the CPM R-CNN hot path (counterpart of pet/rcnn/core/config.py).

Same key names, defaults, YAML/CLI merge and type-coercion rules as the reference for every key the
detection hot path (and the BASELINE YAMLs under cfgs/rcnn/mscoco/grid_cascade/) reads.  Defaults are cited
by reference line.  Keys for tasks outside the hot path (mask / keypoint / parsing / UV / retinanet / FCOS,
visualisation) are intentionally absent; merging a YAML that sets one raises KeyError as the reference does
for unknown keys.  YAML is read with yaml.safe_load (the reference's bare yaml.load breaks on PyYAML >= 6).
"""
import copy
from ast import literal_eval

import numpy as np
import yaml

from pet.utils.collections import AttrDict


def _tree(d):
    return AttrDict({k: _tree(v) if isinstance(v, dict) else v for k, v in d.items()})


_DEFAULTS = {
    "DEVICE": "cuda", "NUM_GPUS": 1, "DISPLAY_ITER": 20, "CKPT": "",                         # config.py:19-52
    "PIXEL_MEANS": np.array([102.9801, 115.9465, 122.7717]), "PIXEL_STDS": np.array([1.0, 1.0, 1.0]),
    "TO_BGR255": True,                                                                       # :46
    "DATALOADER": {"SAMPLER_TRAIN": "DistributedSampler", "ASPECT_RATIO_GROUPING": True},   # :205-211
    "MODEL": {                                                                               # :55-131
        "TYPE": "generalized_rcnn", "FPN_ON": False, "FASTER_RCNN": True, "GRID_ON": False, "CASCADE_ON": False,
        "MASK_ON": False, "SEMSEG_ON": False, "KEYPOINT_ON": False, "PARSING_ON": False, "UV_ON": False,
        "HIER_ON": False, "RPN_ONLY": False, "RETINANET_ON": False, "FCOS_ON": False, "MSL_ON": False,
        "BATCH_NORM": "freeze", "NUM_CLASSES": -1, "CLS_AGNOSTIC_BBOX_REG": False, "CONV1_RGB2BGR": True,
    },
    "SOLVER": {                                                                              # :138-196
        "OPTIMIZER": "SGD", "BASE_LR": 0.001, "MAX_ITER": 90000, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0005,
        "WEIGHT_DECAY_GN": 0.0, "BIAS_DOUBLE_LR": True, "BIAS_WEIGHT_DECAY": False, "LR_MULTIPLE": 1.0,
        "WARM_UP_ITERS": 500, "WARM_UP_FACTOR": 1.0 / 10.0, "WARM_UP_METHOD": "LINEAR", "LR_POLICY": "STEP",
        "LR_POW": 0.9, "STEPS": [60000, 80000], "GAMMA": 0.1, "LOG_LR_CHANGE_THRESHOLD": 1.1,
        "SNAPSHOT_ITERS": 10000,
    },
    "TRAIN": {                                                                               # :203-290
        "WEIGHTS": "", "DATASETS": (), "SCALES": (600,), "MAX_SIZE": 1000, "SIZE_DIVISIBILITY": 32,
        "BATCH_SIZE": 16, "FREEZE_CONV_BODY": False, "LOADER_THREADS": 4, "AUTO_RESUME": True,
        "BRIGHTNESS": 0.0, "CONTRAST": 0.0, "SATURATION": 0.0, "HUE": 0.0, "LEFT_RIGHT": (),          # :272-278
    },
    "TEST": {                                                                                # :293-335
        "WEIGHTS": "", "DATASETS": (), "SCALE": 600, "MAX_SIZE": 1000, "SIZE_DIVISIBILITY": 32,
        "IMS_PER_GPU": 1, "LOADER_THREADS": 4,
        "SOFT_NMS": {"ENABLED": False, "METHOD": "linear", "SIGMA": 0.5},
        "BBOX_VOTE": {"ENABLED": False, "VOTE_TH": 0.8, "SCORING_METHOD": "ID", "SCORING_METHOD_BETA": 1.0},
        "BBOX_AUG": {"ENABLED": False, "H_FLIP": False, "SCALES": (), "MAX_SIZE": 4000},
    },
    "BACKBONE": {
        "CONV_BODY": "resnet",
        "RESNET": {                                                                          # :438-489
            "LAYERS": (3, 4, 6, 3), "WIDTH": 64, "BOTTLENECK": True, "STRIDE_3X3": False, "USE_3x3x3HEAD": False,
            "AVG_DOWN": False, "USE_GN": False, "USE_AN": False, "USE_WS": False, "USE_ALIGN": False,
            "STAGE_WITH_CONTEXT": ("none", "none", "none", "none"), "CTX_RATIO": 0.0625,
            "STAGE_WITH_CONV": ("normal", "normal", "normal", "normal"), "C5_DILATION": 1, "FREEZE_AT": 2,
        },
        "RESNEXT": {                                                                         # :494-542
            "LAYERS": (3, 4, 6, 3), "C": 32, "WIDTH": 4, "USE_3x3x3HEAD": False, "AVG_DOWN": False,
            "USE_GN": False, "USE_WS": False, "USE_ALIGN": False,
            "STAGE_WITH_CONTEXT": ("none", "none", "none", "none"), "CTX_RATIO": 0.0625,
            "STAGE_WITH_CONV": ("normal", "normal", "normal", "normal"), "C5_DILATION": 1, "FREEZE_AT": 2,
        },
    },
    "FPN": {                                                                                 # :552-603
        "BODY": "fpn", "USE_C5": True, "DIM": 256, "LOWEST_BACKBONE_LVL": 2, "HIGHEST_BACKBONE_LVL": 5,
        "MULTILEVEL_ROIS": True, "ROI_CANONICAL_SCALE": 224, "ROI_CANONICAL_LEVEL": 4, "ROI_MAX_LEVEL": 5,
        "ROI_MIN_LEVEL": 2, "MULTILEVEL_RPN": True, "RPN_MAX_LEVEL": 6, "RPN_MIN_LEVEL": 2,
        "EXTRA_CONV_LEVELS": False, "USE_LITE": False, "USE_BN": False, "USE_GN": False, "USE_WS": False,
    },
    "RPN": {                                                                                 # :675-740
        "ANCHOR_SIZES": (32, 64, 128, 256, 512), "ANCHOR_STRIDE": (16,), "ASPECT_RATIOS": (0.5, 1.0, 2.0),
        "STRADDLE_THRESH": 0, "FG_IOU_THRESHOLD": 0.7, "BG_IOU_THRESHOLD": 0.3, "BATCH_SIZE_PER_IMAGE": 256,
        "POSITIVE_FRACTION": 0.5, "PRE_NMS_TOP_N_TRAIN": 12000, "PRE_NMS_TOP_N_TEST": 6000,
        "POST_NMS_TOP_N_TRAIN": 2000, "POST_NMS_TOP_N_TEST": 1000, "NMS_THRESH": 0.7, "MIN_SIZE": 0,
        "FPN_POST_NMS_TOP_N_TRAIN": 2000, "FPN_POST_NMS_TOP_N_TEST": 2000, "FPN_POST_NMS_PER_BATCH": True,
        "SMOOTH_L1_BETA": 1.0 / 9, "RPN_HEAD": "SingleConvRPNHead",
    },
    "FAST_RCNN": {                                                                           # :744-845 (MLP head subset)
        "ROI_BOX_HEAD": "roi_2mlp_head", "ROI_BOX_OUTPUT": "Box_output", "ROI_XFORM_METHOD": "ROIAlign",
        "ROI_XFORM_SAMPLING_RATIO": 0, "ROI_XFORM_RESOLUTION": (14, 14), "FG_IOU_THRESHOLD": 0.5,
        "BG_IOU_THRESHOLD": 0.5, "BBOX_REG_WEIGHTS": (10., 10., 5., 5.), "BATCH_SIZE_PER_IMAGE": 512,
        "POSITIVE_FRACTION": 0.25, "SCORE_THRESH": 0.05, "NMS": 0.5, "DETECTIONS_PER_IMG": 100, "SMOOTH_L1_BETA": 1,
        "MLP_HEAD": {"MLP_DIM": 1024, "USE_BN": False, "USE_GN": False, "USE_WS": False},
    },
    "CASCADE_RCNN": {                                                                        # :1021-1061
        "ROI_BOX_HEAD": "roi_2mlp_head", "ROI_BOX_OUTPUT": "Box_output", "NUM_STAGE": 3,
        "FG_IOU_THRESHOLD": [0.5, 0.6, 0.7], "BG_IOU_THRESHOLD": [0.5, 0.6, 0.7],
        "BBOX_REG_WEIGHTS": ((10., 10., 5., 5.), (20., 20., 10., 10.), (30., 30., 15., 15.)),
        "STAGE_WEIGHTS": (1.0, 0.5, 0.25), "TEST_STAGE": 3, "TEST_ENSEMBLE": True, "RESCORE_ON": False,
        "IOU_HELPER": False, "IOU_HELPER_MERGE": False, "IOU_LOSS_WEIGHT": 1.0, "RESCORE_LOSS_WEIGHT": 1.0,
    },
    "GRID_RCNN": {                                                                           # :850-1008
        "CASCADE_MAPPING_ON": False, "RESCORE_ON": False, "ROI_GRID_HEAD": "roi_grid_head",
        "ROI_GRID_OUTPUT": "Grid_output", "ROI_CLS_HEAD": "roi_cls_head", "ROI_CLS_OUTPUT": "Cls_output",
        "MAX_SAMPLE_NUM_GRID": 96, "ACROSS_SAMPLE": False, "ROI_XFORM_METHOD": "ROIAlign",
        "ROI_XFORM_SAMPLING_RATIO": 2, "ROI_XFORM_RESOLUTION_CLS": (7, 7), "ROI_XFORM_RESOLUTION_GRID": (14, 14),
        "FG_IOU_THRESHOLD": 0.5, "BG_IOU_THRESHOLD": 0.5, "BATCH_SIZE_PER_IMAGE": 512, "POSITIVE_FRACTION": 0.25,
        "SCORE_THRESH": 0.03, "NMS": 0.3, "LOSS_WEIGHT": 15, "POS_RADIUS": 1, "GRID_POINTS": 9,
        "ROI_FEAT_SIZE": 14, "RANDOM_JITTER": False, "FINEST_LEVEL_ROI": False, "TARGET_REFINE": False,
        "BETTER_ROI": False, "BETTER_ROI_RATIO": 0.25, "ENHANCE_FEATURES": False, "FUSED_ON": True,
        "EXTEND_ROI": False, "OFFSET_ON": False, "IOU_HELPER": False, "IOU_HELPER_MERGE": False,
        "IOU_LOSS_WEIGHT": 1.0, "RESCORE_LOSS_WEIGHT": 1.0, "SE_ON": False,
        "RESCORE_OPTION": {"KEEP_RATIO": False},
        "MLP_HEAD": {"MLP_DIM": 1024, "USE_BN": False, "USE_GN": False, "USE_WS": False},
        "GRID_HEAD": {"NUM_CONVS": 8, "POINT_FEAT_CHANNELS": 64},
        "CASCADE_MAPPING_OPTION": {
            "STAGE_NUM": 3, "TEST_STAGE": 3, "TEST_ENSEMBLE": True, "STAGE_WEIGHTS": (1.0, 0.5, 0.25),
            "STAGE_MAPPING_RATIO": (1.0, 0.5, 0.25), "FG_IOU_THRESHOLD": [0.5, 0.6, 0.7],
            "BG_IOU_THRESHOLD": [0.5, 0.6, 0.7], "GRID_NUM": (9, 9, 9), "RESIZE_ROI": False,
        },
    },
    # Visualisation options (config.py:1143-1276).  Nothing on the hot path reads them; the subtree exists because the
    # repo's YAMLs set VIS keys and a merge of an unknown key raises (as in the reference).
    "VIS": {
        "ENABLED": False, "VIS_TH": 0.9,
        "SHOW_BOX": {"ENABLED": True, "COLOR_SCHEME": "green", "COLORMAP": "COCO81", "BORDER_THICK": 2},
        "SHOW_CLASS": {"ENABLED": True, "COLOR": (218, 227, 218), "FONT_SCALE": 0.45},
        "SHOW_SEGMS": {"ENABLED": True, "SHOW_MASK": True, "MASK_COLOR_FOLLOW_BOX": True, "MASK_ALPHA": 0.4,
                       "SHOW_BORDER": True, "BORDER_COLOR": (255, 255, 255), "BORDER_THICK": 2},
        "SHOW_KPS": {"ENABLED": True, "KPS_TH": 2, "KPS_COLOR_WITH_PARSING": (255, 255, 255), "KPS_ALPHA": 0.7,
                     "LINK_THICK": 2, "CIRCLE_RADIUS": 3, "CIRCLE_THICK": -1},
        "SHOW_PARSS": {"ENABLED": True, "COLORMAP": "CIHP20", "PARSING_ALPHA": 0.4, "SHOW_BORDER": True,
                       "BORDER_COLOR": (255, 255, 255), "BORDER_THICK": 1},
        "SHOW_UV": {"ENABLED": True, "SHOW_BORDER": True, "BORDER_THICK": 6, "GRID_THICK": 2, "LINES_NUM": 15},
    },
}

__C = _tree(_DEFAULTS)
cfg = __C

_RENAMED_KEYS = {"PIXEL_MEAN": "PIXEL_MEANS", "PIXEL_STD": "PIXEL_STDS"}                    # config.py:1293-1297


def reset_cfg():
    """Restore the defaults in place (tests build several configurations in one process)."""
    fresh = _tree(_DEFAULTS)
    cfg.immutable(False)
    cfg.clear()
    cfg.update(fresh)


def assert_and_infer_cfg(make_immutable=True):
    if make_immutable:
        cfg.immutable(True)


def merge_cfg_from_file(cfg_filename):
    with open(cfg_filename, "r") as f:
        loaded = yaml.safe_load(f) or {}
    _merge_a_into_b(AttrDict(loaded), __C)


def merge_cfg_from_list(cfg_list):
    """`['TEST.NMS', 0.5, ...]` style overrides (config.py:1328-1349)."""
    assert len(cfg_list) % 2 == 0
    for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
        if full_key in _RENAMED_KEYS:
            raise KeyError("Key {} was renamed to {}; please update your config.".format(full_key,
                                                                                        _RENAMED_KEYS[full_key]))
        node = __C
        parts = full_key.split(".")
        for sub in parts[:-1]:
            assert sub in node, "Non-existent key: {}".format(full_key)
            node = node[sub]
        assert parts[-1] in node, "Non-existent key: {}".format(full_key)
        node[parts[-1]] = _coerce(_decode(v), node[parts[-1]], full_key)


def _merge_a_into_b(a, b, stack=()):
    for k, raw in a.items():
        full_key = ".".join(stack + (k,))
        if k not in b:
            if full_key in _RENAMED_KEYS:
                raise KeyError("Key {} was renamed to {}".format(full_key, _RENAMED_KEYS[full_key]))
            raise KeyError("Non-existent config key: {}".format(full_key))
        v = _decode(copy.deepcopy(raw))
        if isinstance(v, dict):
            if not isinstance(b[k], AttrDict):
                raise ValueError("Type mismatch for config key: {}".format(full_key))
            _merge_a_into_b(AttrDict(v), b[k], stack + (k,))
        else:
            b[k] = _coerce(v, b[k], full_key)


def _decode(v):
    if isinstance(v, dict):
        return AttrDict(v)
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, full_key):
    """Type rule of config.py:1409-1437: exact type, or ndarray / str / tuple<->list coercions."""
    if type(new) is type(old):
        return new
    if isinstance(old, np.ndarray):
        return np.array(new, dtype=old.dtype)
    if isinstance(old, str):
        return str(new)
    if isinstance(new, tuple) and isinstance(old, list):
        return list(new)
    if isinstance(new, list) and isinstance(old, tuple):
        return tuple(new)
    raise ValueError("Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}".format(
        type(old), type(new), old, new, full_key))
