"""PoolPointsInterp operator surface (counterpart of pet/lib/ops/pool_points_interp.py:10-53)."""
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _C


class _PoolPointsInterp(Function):
    @staticmethod
    def forward(ctx, input, roi, spatial_scale):
        ctx.save_for_backward(roi)
        ctx.spatial_scale = spatial_scale
        ctx.input_shape = input.size()
        return _C.pool_points_interp_forward(input.float(), roi.float(), spatial_scale)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        rois, = ctx.saved_tensors
        bs, ch, h, w = ctx.input_shape
        return _C.pool_points_interp_backward(grad_output, rois, ctx.spatial_scale, bs, ch, h, w), None, None


pool_points_interp = _PoolPointsInterp.apply


class PoolPointsInterp(nn.Module):
    def __init__(self, spatial_scale=1.0):
        super().__init__()
        self.spatial_scale = spatial_scale

    def forward(self, input, rois):
        return pool_points_interp(input, rois, self.spatial_scale)

    def __repr__(self):
        return "{}(spatial_scale={})".format(self.__class__.__name__, self.spatial_scale)
