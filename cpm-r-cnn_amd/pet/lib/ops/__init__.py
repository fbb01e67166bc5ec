"""Operator surface (counterpart of pet/lib/ops/__init__.py:1-30): the hot-path operators over the HIP kernels, plus
every other name of the reference package as an importable off-path placeholder (offpath.py)."""
from .nms import nms, ml_nms, nms_segments, soft_nms, ml_soft_nms, soft_nms_segments
from .roi_align import roi_align, ROIAlign
from .pooler_fpn import roi_align_fpn, roi_backward_group
from .affine import AffineChannel2d
from .losses import smooth_l1_loss, l2_loss, l2_loss_nosync, l2_loss_fused, cross_entropy_fused
from .modules import Conv2d, Linear, ConvTranspose2d, GroupNorm, ReLU
from .conv import conv2d, linear, conv_transpose2d, group_norm, stem_forward, mark_shared_grad, rpn_predictors
from .conv import fwd_fork, fwd_side, fwd_join, rpn_head, set_rpn_sample, mask_compact
from .pool_points_interp import pool_points_interp, PoolPointsInterp
from .boxes import box_iou, box_voting, box_ml_voting
from .deform_conv import deform_conv, cols_conv, DeformConv, DeformConvPack
from .detect_glue import match_rois, grid_bce_loss, grid_decode, rpn_decode, topk_rows, topk_rows_multi, rpn_loss, sample_pos_neg
from .detect_glue import sigmoid_multi, rpn_decode_multi, rpn_labels
from .image_prep import image_prep, resample_tables, value_table, resize_linear
from .offpath import *  # noqa: F401,F403  (every remaining name of the reference's ops package imports)
