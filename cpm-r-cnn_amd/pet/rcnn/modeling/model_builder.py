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

    # ---- the static part of the training step as hipGraphs --------------------------------------------------------------
    # Backbone, FPN and the RPN head's convolutions have no data-dependent shape: for a given padded batch size they are
    # the same ~370 kernel launches forward and ~600 backward every step, issued by ~130 Python autograd nodes.  On a
    # busy host that Python work, not the device, bounds the step.  capture_static_part() records both directions once
    # (torch.cuda.make_graphed_callables: warm-up, then a forward and a backward hipGraph in a private memory pool, the
    # second stream's weight gradients included through the fork / join events) and forward() replays them for batches of
    # the captured shape; the weight gradients still land in the flat optimizer's buffer (the kernels' sinks are fixed
    # addresses), other shapes run eagerly.  What a replay cannot do is call back into Python: the data-parallel
    # reducer hears about these parameters at FlatGradReducer.finish() instead of during the backward pass.
    def _static_part(self):
        model = self

        class StaticPart(nn.Module):
            def forward(self, x):
                feats = model._features(x)
                obj, reg = model.RPN.head(feats)
                return tuple(feats) + tuple(obj) + tuple(reg)
        return StaticPart()

    def capture_static_part(self, sample):
        """sample: an image batch tensor [N,3,H,W] of the shape to capture (its values are used for the warm-up only).
        Call it once the optimizer has stepped at least once (the data-gradient weight images exist from then on) and
        BEFORE zero_grad of the next step: warm-up and capture leave gradients in the parameters' sinks."""
        import torch
        if not self.training:
            raise RuntimeError("capture_static_part is for training steps")
        part = self._static_part()
        params = [p for p in list(self.Conv_Body.parameters()) + list(self.RPN.head.parameters()) +
                  (list(self.Conv_Body_FPN.parameters()) if cfg.MODEL.FPN_ON else []) if p.requires_grad]
        # make_graphed_callables finds a module's parameters through .parameters(): hand it the real ones
        for i, p_ in enumerate(params):
            part.register_parameter("p%d" % i, p_)
        x = sample.detach().clone()
        from pet.lib.ops import conv as _conv
        _conv.wait_pending_wt(x.device)          # (the capture must not record a wait for an event of the eager step)
        from pet.lib.ops import _hip as _H
        _H.wait_pending_sgd(x.device)            # (nor for an optimizer update queued beside the next forward pass)
        graphed = torch.cuda.make_graphed_callables(part, (x,), allow_unused_input=True)
        self._graphed = getattr(self, "_graphed", {})
        self._graphed[tuple(sample.shape)] = graphed
        return graphed

    # ---- the static part of the TEST-time forward as a hipGraph --------------------------------------------------------
    # One image per forward (the reference's inference): ~100 convolutions of backbone, FPN and RPN head issued by Python
    # take the host 4-5 ms, the device ~2.  For every padded input shape seen (up to CPM_EVAL_GRAPH_SHAPES, 8) the static
    # part is captured once -- two eager warm-up runs, then a capture on the same tensors -- and replayed afterwards: the
    # input is copied into the captured buffer, the outputs are the captured buffers (valid until the next replay of
    # that shape; box_net, whose caller keeps the features across passes, returns copies).  The captured kernels read parameters through their
    # addresses, but the bf16x3 weight images of parameters OUTSIDE a flat optimizer are re-split by Python when a
    # parameter changes (ops.conv.w4_of): a graph is therefore keyed by the parameters' version counters too and
    # re-captured when they move.  CPM_EVAL_GRAPH=0: eager.
    def _eval_static(self, x):
        import os
        import torch
        if os.environ.get("CPM_EVAL_GRAPH", "1") == "0" or not x.is_cuda or torch.is_grad_enabled() \
                or torch.cuda.is_current_stream_capturing():
            return None
        from pet.lib.ops import _hip as _H, conv as _conv
        graphs = self.__dict__.setdefault("_eval_graphs", {})
        params = self.__dict__.get("_eval_graph_params")
        if params is None:
            params = self.__dict__["_eval_graph_params"] = [
                p for p in list(self.Conv_Body.parameters()) + list(self.RPN.head.parameters()) +
                (list(self.Conv_Body_FPN.parameters()) if cfg.MODEL.FPN_ON else [])]
        epoch = 0
        for p_ in params:
            epoch += p_._version
        key = (tuple(x.shape), x.device.index, _H.get_conv_math(), _H.deterministic(), x.is_contiguous())
        rec = graphs.get(key)
        if rec is not None and rec[0] != epoch:
            rec = None
        if rec is False:
            return None
        if rec is None:
            if len(graphs) >= int(os.environ.get("CPM_EVAL_GRAPH_SHAPES", "8")) and key not in graphs:
                return None
            part = self._static_part()
            static_in = x.detach().clone()
            try:
                _conv.wait_pending_wt(x.device)
                _H.wait_pending_sgd(x.device)
                for _ in range(2):                       # warm-up: weight images, plans, workspaces, allocator blocks
                    part(static_in)
                torch.cuda.current_stream(x.device).synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    outs = part(static_in)
            except Exception as e:                       # (a capture that cannot be made: stay eager for this shape)
                import warnings
                warnings.warn("test-time hipGraph of the static part not captured for %s: %s" % (key[0], e))
                graphs[key] = False
                return None
            rec = graphs[key] = (epoch, g, static_in, outs)
        _, g, static_in, outs = rec
        static_in.copy_(x)
        g.replay()
        self.__dict__["_eval_graphs_last"] = outs
        return outs

    def forward(self, images, targets=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        images = to_image_list(images)
        graphed = getattr(self, "_graphed", {}).get(tuple(images.tensors.shape)) if self.training else None
        if graphed is not None:
            # a replayed backward pass issues no Python-side waits: the once-per-step weight-image transform on the
            # second stream (FlatSGD._refresh_dgrad_weights) is ordered in front of the replay here
            from pet.lib.ops import _hip as _H, conv as _conv
            _conv.wait_pending_wt(images.tensors.device)
            _H.wait_pending_sgd(images.tensors.device)        # (an optimizer update queued beside this forward pass)
            outs = graphed(images.tensors)
            nf = nl = len(outs) // 3                 # feature maps, objectness maps, delta maps: one each per level
            # the replayed outputs are fresh tensor objects: the RoI heads' RoIAlign calls share ONE gradient
            # accumulator per level again (ops.mark_shared_grad, as FPN.forward does for the eager tensors)
            feats = [ops.mark_shared_grad(o) for o in outs[:nf]]
            proposals, proposal_losses = self.RPN(images, feats, targets,
                                                  head_out=(list(outs[nf:nf + nl]), list(outs[nf + nl:])))
        elif not self.training and not cfg.MODEL.RPN_ONLY and self._eval_static(images.tensors) is not None:
            outs = self._eval_graphs_last
            nf = nl = len(outs) // 3
            feats = list(outs[:nf])
            proposals, proposal_losses = self.RPN(images, feats, targets,
                                                  head_out=(list(outs[nf:nf + nl]), list(outs[nf + nl:])))
        else:
            feats = self._features(images.tensors)
            if feats and feats[0].is_cuda:
                # FlatSGD.overlap_next_forward: the first op on a trainable tensor has waited for the update already
                # (H.require_gpu); this is the net under a body without one
                from pet.lib.ops import _hip as _H
                _H.wait_pending_sgd(feats[0].device)
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
        x = self.Norm(images.tensors)
        outs = None if (self.training or cfg.MODEL.RPN_ONLY) else self._eval_static(x)
        if outs is not None:
            # the caller keeps the features across passes (test-time augmentation: core/test.py collects them per
            # pass): copies, not the graph's buffers
            nf = nl = len(outs) // 3
            feats = [o.clone() for o in outs[:nf]]
            proposals, _ = self.RPN(images, feats, targets, head_out=(list(outs[nf:nf + nl]), list(outs[nf + nl:])))
        else:
            feats = self._features(x)
            proposals, _ = self.RPN(images, feats, targets)
        if not cfg.MODEL.RPN_ONLY:
            _, result, _ = self._roi_heads()(feats, proposals, targets)
        else:
            result = proposals
        return feats, result
