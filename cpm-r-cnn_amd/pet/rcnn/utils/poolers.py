"""FPN RoI pooler (counterpart of pet/rcnn/utils/poolers.py:9-132).

The reference maps RoIs to levels with torch ops, then loops over levels (nonzero -> gather -> RoIAlign ->
scatter).  Here a multi-level Pooler is ONE fused HIP launch (cpm_roi_align_fpn_*): the level of each RoI is
computed in-kernel with the LevelMapper formula and results land in RoI order."""
import torch
from torch import nn

from pet.lib.ops import ROIAlign
from pet.lib.ops.pooler_fpn import roi_align_fpn
from pet.rcnn.utils.misc import cat


class LevelMapper(object):
    """floor(lvl0 + log2(sqrt(area)/s0 + eps)) clamped to [k_min, k_max], minus k_min (poolers.py:30-40).
    Kept for API parity and for host-side checks; the fused pooler evaluates the same formula on the device."""

    def __init__(self, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
        self.k_min, self.k_max, self.s0, self.lvl0, self.eps = k_min, k_max, canonical_scale, canonical_level, eps

    def __call__(self, boxlists):
        s = torch.sqrt(cat([b.area() for b in boxlists]))
        lvls = torch.floor(self.lvl0 + torch.log2(s / self.s0 + self.eps))
        return torch.clamp(lvls, min=self.k_min, max=self.k_max).to(torch.int64) - self.k_min


class Pooler(nn.Module):
    def __init__(self, method, output_size, scales, sampling_ratio, rotated=False, interpolation="bilinear"):
        assert method in {"ROIAlign", "ROIAlignV2"}, "only ROIAlign is on the CPM R-CNN hot path, got {}".format(method)
        assert not rotated and interpolation == "bilinear"
        super().__init__()
        self.aligned = "V2" in method
        self.scales = [float(s) for s in scales]
        self.sampling_ratio = sampling_ratio
        self.output_size = tuple(output_size)
        self.poolers = nn.ModuleList([ROIAlign(output_size, spatial_scale=s, sampling_ratio=sampling_ratio,
                                               aligned=self.aligned) for s in scales])
        lvl_min = -torch.log2(torch.tensor(scales[0], dtype=torch.float32)).item()
        lvl_max = -torch.log2(torch.tensor(scales[-1], dtype=torch.float32)).item()
        self.map_levels = LevelMapper(lvl_min, lvl_max)

    @staticmethod
    def convert_to_roi_format(boxes):
        concat = cat([b.bbox for b in boxes], dim=0)
        ids = cat([torch.full((len(b), 1), i, dtype=concat.dtype, device=concat.device)
                   for i, b in enumerate(boxes)], dim=0)
        return torch.cat([ids, concat], dim=1)

    def forward(self, x, boxes):
        # lists that come packed from the device (pet/lib/ops/roi_lists.py) carry their [R, 5] rows ready-made
        rois = getattr(boxes, "rois5", None)
        if rois is None:
            rois = self.convert_to_roi_format(boxes)
        if len(self.poolers) == 1:
            return self.poolers[0](x[0], rois)
        assert not self.aligned
        return roi_align_fpn(list(x[:len(self.poolers)]), rois, self.output_size, self.scales, self.sampling_ratio,
                             self.map_levels.s0, self.map_levels.lvl0, self.map_levels.eps)
