"""build_transforms (pet/rcnn/datasets/transform.py:6-50): the same chain, on deferred images."""
from pet.rcnn.core.config import cfg
from pet.utils.data import transforms as T


def build_transforms(is_train=True):
    if is_train:
        min_size, max_size, flip_prob = cfg.TRAIN.SCALES, cfg.TRAIN.MAX_SIZE, 0.5
        jitter = (cfg.TRAIN.BRIGHTNESS, cfg.TRAIN.CONTRAST, cfg.TRAIN.SATURATION, cfg.TRAIN.HUE)
        left_right = cfg.TRAIN.LEFT_RIGHT
    else:
        min_size, max_size, flip_prob = cfg.TEST.SCALE, cfg.TEST.MAX_SIZE, 0
        jitter = (0.0, 0.0, 0.0, 0.0)
        left_right = ()
    return T.Compose([
        T.ColorJitter(*jitter),
        T.Resize(min_size, max_size),
        T.RandomHorizontalFlip(flip_prob, left_right),
        T.ToTensor(),
        T.Normalize(mean=cfg.PIXEL_MEANS, std=cfg.PIXEL_STDS, to_bgr255=cfg.TO_BGR255),
    ])
