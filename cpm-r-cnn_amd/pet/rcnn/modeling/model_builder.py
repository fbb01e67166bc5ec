"""Generalized_RCNN (counterpart of pet/rcnn/modeling/model_builder.py:19-195), CPM R-CNN wiring:
Conv_Body -> Conv_Body_FPN -> RPN -> Grid_Cascade_RCNN (MODEL.GRID_ON) or Cascade_RCNN (MODEL.FASTER_RCNN +
MODEL.CASCADE_ON, the offset-regression cascade with ISM / RSM).  Attribute names are the state-dict prefixes of the
released checkpoints.  Images enter as NCHW fp32 (the loader's layout); everything downstream is NHWC."""
import numpy as np
import torch
import torch.nn as nn

import pet.lib.ops as ops
import pet.rcnn.modeling.backbone  # noqa: F401  (registers the bodies)
import pet.rcnn.modeling.fpn  # noqa: F401
from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry
from pet.rcnn.modeling.cascade_rcnn.cascade_rcnn import CascadeRCNN
from pet.rcnn.modeling.grid_cascade_rcnn.grid_cascade_rcnn import GridCascadeRCNN
from pet.rcnn.modeling.rpn.rpn import build_rpn
from pet.utils.data.structures.image_list import to_image_list


class Generalized_RCNN(nn.Module):
    def __init__(self, is_train=True):
        super().__init__()
        M = cfg.MODEL
        if M.MASK_ON or M.KEYPOINT_ON or M.PARSING_ON or M.UV_ON or M.SEMSEG_ON or M.RETINANET_ON or M.FCOS_ON:
            raise ValueError("only the box-detection path is built (mask/keypoint/parsing/uv/semseg heads are out of scope)")
        if not is_train:
            self.Norm = ops.AffineChannel2d(3)
            self.Norm.weight.data = torch.from_numpy(1. / np.array(cfg.PIXEL_STDS)).float()
            self.Norm.bias.data = torch.from_numpy(-1. * np.array(cfg.PIXEL_MEANS) / np.array(cfg.PIXEL_STDS)).float()
        self.Conv_Body = registry.BACKBONES[cfg.BACKBONE.CONV_BODY]()
        self.dim_in = self.Conv_Body.dim_out
        self.spatial_scale = self.Conv_Body.spatial_scale
        if M.FPN_ON:
            self.Conv_Body_FPN = registry.FPN_BODY[cfg.FPN.BODY](self.dim_in, self.spatial_scale)
            self.dim_in = self.Conv_Body_FPN.dim_out
            self.spatial_scale = self.Conv_Body_FPN.spatial_scale
        else:
            self.dim_in = self.dim_in[-1:]
            self.spatial_scale = self.spatial_scale[-1:]
        self.RPN = build_rpn(self.dim_in)
        if not M.RPN_ONLY:
            if M.FASTER_RCNN:
                if not M.CASCADE_ON:
                    raise ValueError("single-stage Fast R-CNN heads are not built (MODEL.CASCADE_ON or MODEL.GRID_ON)")
                self.Cascade_RCNN = CascadeRCNN(self.dim_in, self.spatial_scale)
            elif M.GRID_ON and cfg.GRID_RCNN.CASCADE_MAPPING_ON:
                self.Grid_Cascade_RCNN = GridCascadeRCNN(self.dim_in, self.spatial_scale)
            else:
                raise ValueError("built RoI heads: MODEL.GRID_ON + GRID_RCNN.CASCADE_MAPPING_ON (CPM R-CNN) and "
                                 "MODEL.FASTER_RCNN + MODEL.CASCADE_ON (Cascade R-CNN)")
        if not M.RPN_ONLY:
            # heads that consume the proposals as a packed device list let the RPN skip its host round trips
            self.RPN.roi_heads_take_lists = bool(getattr(self._roi_heads(), "takes_device_lists", False))
        if cfg.TRAIN.FREEZE_CONV_BODY:
            for p in self.Conv_Body.parameters():
                p.requires_grad = False
            if M.FPN_ON:
                for p in self.Conv_Body_FPN.parameters():
                    p.requires_grad = False

    def _roi_heads(self):
        return self.Cascade_RCNN if cfg.MODEL.FASTER_RCNN else self.Grid_Cascade_RCNN

    def _features(self, x):
        feats = self.Conv_Body(x)
        return self.Conv_Body_FPN(feats) if cfg.MODEL.FPN_ON else [feats[-1]]

    def forward(self, images, targets=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        images = to_image_list(images)
        feats = self._features(images.tensors)
        proposals, proposal_losses = self.RPN(images, feats, targets)
        roi_losses = {}
        if not cfg.MODEL.RPN_ONLY:
            _, result, roi_losses = self._roi_heads()(feats, proposals, targets)
        else:
            result = proposals
        if self.training:
            losses = {}
            losses.update(proposal_losses)
            losses.update(roi_losses)
            return {"metrics": {}, "losses": losses}
        return result

    def box_net(self, images, targets=None):
        images = to_image_list(images, cfg.TEST.SIZE_DIVISIBILITY)
        feats = self._features(self.Norm(images.tensors))
        proposals, _ = self.RPN(images, feats, targets)
        if not cfg.MODEL.RPN_ONLY:
            _, result, _ = self._roi_heads()(feats, proposals, targets)
        else:
            result = proposals
        return feats, result
