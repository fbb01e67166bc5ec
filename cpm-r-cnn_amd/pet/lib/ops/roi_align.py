"""RoIAlign operator surface (reference: pet/lib/ops/roi_align.py:14-95)."""
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import _C

INTERPOLATION_METHOD = {"bilinear": 0, "nearest": 1}


class _ROIAlign(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio, aligned, interpolation="bilinear"):
        ctx.save_for_backward(roi)
        ctx.geom = (_pair(output_size), float(spatial_scale), int(sampling_ratio), bool(aligned),
                    INTERPOLATION_METHOD[interpolation], tuple(input.shape))
        oh, ow = ctx.geom[0]
        return _C.roi_align_forward(input.float(), roi.float(), spatial_scale, oh, ow, sampling_ratio, aligned,
                                    ctx.geom[4])

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        roi, = ctx.saved_tensors
        (oh, ow), scale, ratio, aligned, interp, (bs, ch, h, w) = ctx.geom
        grad_input = _C.roi_align_backward(grad_output, roi.float(), scale, oh, ow, bs, ch, h, w, ratio, aligned,
                                           interp)
        return grad_input, None, None, None, None, None, None


roi_align = _ROIAlign.apply


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned, interpolation="bilinear"):
        assert interpolation in INTERPOLATION_METHOD, "Unknown interpolation method: {}".format(interpolation)
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.aligned = aligned
        self.interpolation_method = interpolation

    def forward(self, input, rois):
        """input: [N,C,H,W] (either memory format); rois: [K,5] = (batch index, x1, y1, x2, y2)."""
        assert rois.dim() == 2 and rois.size(1) == 5
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned,
                         self.interpolation_method)

    def __repr__(self):
        return "{}(output_size={}, spatial_scale={}, sampling_ratio={}, aligned={})".format(
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)
