"""RPN anchors (counterpart of pet/rcnn/modeling/rpn/anchor_generator.py:34-125,221-290).

Cell anchors follow the Faster R-CNN enumeration (ratio first, then scale) in float64 and are stored as fp32
buffers named cell_anchors.N (state-dict ABI).  Per-level grid anchors are cached per feature-map size: they are
pure functions of the shapes, so a training run computes them once instead of every step."""
import numpy as np
import torch
from torch import nn

from pet.rcnn.core.config import cfg
from pet.utils.data.structures.bounding_box import BoxList


class BufferList(nn.Module):
    def __init__(self, buffers=None):
        super().__init__()
        for i, b in enumerate(buffers or []):
            self.register_buffer(str(i), b)

    def __len__(self):
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


def _centered(ws, hs, xc, yc):
    ws, hs = ws[:, None], hs[:, None]
    return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))


def _whc(a):
    w, h = a[2] - a[0] + 1, a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    """Anchors around a (0,0,stride-1,stride-1) window: rounded ratio enumeration, then scales = size/stride."""
    scales = np.array(sizes, dtype=np.float64) / stride
    ratios = np.array(aspect_ratios, dtype=np.float64)
    base = np.array([1, 1, stride, stride], dtype=np.float64) - 1
    w, h, xc, yc = _whc(base)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    per_ratio = _centered(ws, hs, xc, yc)
    out = []
    for a in per_ratio:
        w, h, xc, yc = _whc(a)
        out.append(_centered(w * scales, h * scales, xc, yc))
    return torch.from_numpy(np.vstack(out))


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=(128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0), anchor_strides=(8, 16, 32),
                 straddle_thresh=0):
        super().__init__()
        if len(anchor_strides) == 1:
            cells = [generate_anchors(anchor_strides[0], sizes, aspect_ratios).float()]
        else:
            if len(anchor_strides) != len(sizes):
                raise RuntimeError("FPN should have #anchor_strides == #sizes")
            cells = [generate_anchors(st, sz if isinstance(sz, (tuple, list)) else (sz,), aspect_ratios).float()
                     for st, sz in zip(anchor_strides, sizes)]
        self.strides = anchor_strides
        self.cell_anchors = BufferList(cells)
        self.straddle_thresh = straddle_thresh
        self._cache = {}
        self._image_cache, self._batch_cache = {}, {}

    def num_anchors_per_location(self):
        return [len(c) for c in self.cell_anchors]

    def grid_anchors(self, grid_sizes):
        out = []
        for (gh, gw), stride, base in zip(grid_sizes, self.strides, self.cell_anchors):
            key = (int(gh), int(gw), int(stride), str(base.device))
            if key not in self._cache:
                sx = torch.arange(0, gw * stride, step=stride, dtype=torch.float32, device=base.device)
                sy = torch.arange(0, gh * stride, step=stride, dtype=torch.float32, device=base.device)
                yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                xx, yy = xx.reshape(-1), yy.reshape(-1)
                shifts = torch.stack((xx, yy, xx, yy), dim=1)
                self._cache[key] = (shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)
            out.append(self._cache[key])
        return out

    def visibility(self, anchors, image_width, image_height):
        if self.straddle_thresh >= 0:
            t = self.straddle_thresh
            return ((anchors[..., 0] >= -t) & (anchors[..., 1] >= -t) & (anchors[..., 2] < image_width + t)
                    & (anchors[..., 3] < image_height + t))
        return torch.ones(anchors.shape[0], dtype=torch.bool, device=anchors.device)

    def add_visibility_to(self, boxlist):
        w, h = boxlist.size
        boxlist.add_field("visibility", self.visibility(boxlist.bbox, w, h))

    def forward(self, image_list, feature_maps):
        """list (image) of list (level) of BoxLists with a "visibility" field.  Pure functions of the feature-map and
        image sizes: built once per distinct size and handed out again (read-only) on later steps -- a training step
        otherwise spends ~70 small launches per batch on the visibility masks alone."""
        grid_sizes = tuple((int(f.shape[-2]), int(f.shape[-1])) for f in feature_maps)
        dev = str(feature_maps[0].device)
        per_level = None
        anchors = AnchorBatch()
        for (ih, iw) in image_list.image_sizes:
            key = (grid_sizes, int(ih), int(iw), dev)
            in_image = self._image_cache.get(key)
            if in_image is None:
                if per_level is None:
                    per_level = self.grid_anchors(grid_sizes)
                in_image = []
                for a in per_level:
                    bl = BoxList(a, (iw, ih), mode="xyxy")
                    self.add_visibility_to(bl)
                    in_image.append(bl)
                if len(self._image_cache) >= 256:
                    self._image_cache.clear()
                self._image_cache[key] = in_image
            anchors.append(in_image)
        anchors.key = (grid_sizes, tuple((int(h), int(w)) for h, w in image_list.image_sizes), dev)
        anchors.owner = self
        return anchors

    def batch_pack(self, anchors):
        """(boxes [T, 4], visibility [T], image index int32 [T], anchors per image) of all anchors of the batch,
        image-major then level -- what the batch-fused RPN loss consumes; remembered per (feature, image) sizes."""
        pack = self._batch_cache.get(anchors.key)
        if pack is None:
            n_img = len(anchors)
            abox = torch.cat([a.bbox for per_img in anchors for a in per_img], dim=0)
            vis = torch.cat([a.get_field("visibility") for per_img in anchors for a in per_img], dim=0)
            per = abox.shape[0] // n_img
            img = torch.arange(n_img, device=abox.device).repeat_interleave(per).to(torch.int32)
            if len(self._batch_cache) >= 64:
                self._batch_cache.clear()
            pack = self._batch_cache[anchors.key] = (abox, vis, img, per)
        return pack


class AnchorBatch(list):
    """the anchors of a batch + the key of their cached concatenation (AnchorGenerator.batch_pack)"""
    key = None
    owner = None


def make_anchor_generator():
    R = cfg.RPN
    if cfg.MODEL.FPN_ON:
        assert len(R.ANCHOR_STRIDE) == len(R.ANCHOR_SIZES), "FPN should have len(ANCHOR_STRIDE) == len(ANCHOR_SIZES)"
    else:
        assert len(R.ANCHOR_STRIDE) == 1, "Non-FPN should have a single ANCHOR_STRIDE"
    return AnchorGenerator(R.ANCHOR_SIZES, R.ASPECT_RATIOS, R.ANCHOR_STRIDE, R.STRADDLE_THRESH)
