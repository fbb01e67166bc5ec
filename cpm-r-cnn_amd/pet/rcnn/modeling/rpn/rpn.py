"""RPN module (counterpart of pet/rcnn/modeling/rpn/rpn.py:12-136): shared 3x3 conv (+ReLU fused) and the
1x1 objectness / box-delta convs on every FPN level, proposal selection under no_grad, loss."""
import torch
from torch import nn

import pet.lib.ops as ops
from pet.rcnn.core.config import cfg
from pet.rcnn.modeling.rpn.anchor_generator import make_anchor_generator
from pet.rcnn.modeling.rpn.inference import make_rpn_postprocessor
from pet.rcnn.modeling.rpn.loss import make_rpn_loss_evaluator
from pet.rcnn.utils.box_coder import BoxCoder


import os

_LOSS_ON_SIDE = os.environ.get("CPM_RPN_LOSS_SIDE", "0") != "0"


class RPNHead(nn.Module):
    def __init__(self, dim_in, num_anchors):
        super().__init__()
        self.dim_in = dim_in[-1]
        self.conv = ops.Conv2d(self.dim_in, self.dim_in, kernel_size=3, stride=1, padding=1)
        self.cls_logits = ops.Conv2d(self.dim_in, num_anchors, kernel_size=1, stride=1)
        self.bbox_pred = ops.Conv2d(self.dim_in, num_anchors * 4, kernel_size=1, stride=1)
        for l in (self.conv, self.cls_logits, self.bbox_pred):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)

    def forward(self, x, sparse_backward=False):
        # sparse_backward: the caller's loss is a sum over a SAMPLE of the anchors and announces it (RPNLossComputation
        # -> ops.set_rpn_sample): the whole head is one node whose backward pass touches those anchors only
        if sparse_backward:
            out = ops.rpn_head(list(x), self.conv.weight, self.conv.bias, self.cls_logits.weight, self.cls_logits.bias,
                               self.bbox_pred.weight, self.bbox_pred.bias)
            if out is not None:
                return out
        # both predictors on every level as one autograd node: their data gradients (reductions of 3 and 12, two passes
        # over each level's map) become one launch that also applies this conv's ReLU gate (ops.conv._RPNPredFn)
        # the shared conv on the coarser levels runs on the second stream beside the finest level's (ops.fwd_fork)
        ts = []
        forked = len(x) > 1 and ops.fwd_fork(x[0], 4)
        for li, feature in enumerate(x):
            if forked and li > 0:
                with ops.fwd_side(feature):
                    ts.append(self.conv(feature, relu=True, gate_by_consumers=True))
            else:
                ts.append(self.conv(feature, relu=True, gate_by_consumers=True))
        if forked:
            ops.fwd_join(x[0])
        fused = ops.rpn_predictors(ts, self.cls_logits.weight, self.cls_logits.bias, self.bbox_pred.weight,
                                   self.bbox_pred.bias) if ts else None
        if fused is not None:
            return fused
        logits, bbox_reg = [], []
        for t in ts:
            t = ops.mark_shared_grad(t)
            logits.append(self.cls_logits(t))
            bbox_reg.append(self.bbox_pred(t))
        return logits, bbox_reg


class RPNModule(nn.Module):
    def __init__(self, dim_in):
        super().__init__()
        self.anchor_generator = make_anchor_generator()
        self.head = RPNHead(dim_in, self.anchor_generator.num_anchors_per_location()[0])
        coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.box_selector_train = make_rpn_postprocessor(coder, is_train=True)
        self.box_selector_test = make_rpn_postprocessor(coder, is_train=False)
        self.loss_evaluator = make_rpn_loss_evaluator(coder)

    def forward(self, images, features, targets=None, head_out=None):
        """head_out: (objectness, box deltas) already computed by the caller -- the statically captured part of the step
        (Generalized_RCNN.capture_static_part) ends behind the head's convolutions."""
        # (training under the fused loss: that loss registers its anchor sample, see RPNHead.forward)
        sparse = (self.training and head_out is None and not cfg.MODEL.RPN_ONLY
                  and getattr(self.loss_evaluator, "fused_glue", False))
        objectness, rpn_box_regression = self.head(features, sparse_backward=sparse) if head_out is None else head_out
        anchors = self.anchor_generator(images, features)
        if self.training:
            return self._forward_train(anchors, objectness, rpn_box_regression, targets)
        return self._forward_test(anchors, objectness, rpn_box_regression)

    def _forward_train(self, anchors, objectness, rpn_box_regression, targets):
        sel = self.box_selector_train
        if cfg.MODEL.RPN_ONLY:
            boxes = anchors
        elif sel.can_fuse(objectness):
            # proposals up to the NMS, then the (host-sync-free) loss kernels, then the rest of the selection: the
            # loss runs on the device while the host waits for the NMS counts and builds the proposal lists
            on_device = sel.can_keep_on_device(objectness, targets) and getattr(self, "roi_heads_take_lists", False)
            # The proposal chain (sigmoid, top-k, decode, NMS, finalize: ~0.5 ms of one-to-ten-workgroup kernels) and the
            # loss chain (anchor matching over 2 x 268 569 anchors, labels, sampler, loss + gradients, sample list:
            # ~0.2 ms) both start from the head's outputs and meet nowhere in the forward pass: with the proposals kept
            # on the device the loss chain runs on the second stream beside them.  A real torch stream here (the chain
            # mixes package and framework ops): its autograd nodes belong to that stream, the engine orders their
            # backward against the compute stream by itself.
            side = ops.conv.wgrad_stream(objectness[0].device) if (on_device and _LOSS_ON_SIDE
                                                                    and ops.conv._FWD_SIDE) else None
            if side is not None:
                main = torch.cuda.current_stream(objectness[0].device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    loss_objectness, loss_rpn_box_reg = self.loss_evaluator(anchors, objectness, rpn_box_regression,
                                                                            targets)
            with torch.no_grad():
                pending = sel.start_fused(anchors, objectness, rpn_box_regression, read_counts=not on_device)
                if on_device:
                    boxes = sel.finish_device(pending, targets)         # a packed RoIList; nothing is read back
            if side is None:
                loss_objectness, loss_rpn_box_reg = self.loss_evaluator(anchors, objectness, rpn_box_regression, targets)
            else:
                main.wait_stream(side)        # (cheap: the chain is long finished; keeps every later consumer simple)
            if not on_device:
                with torch.no_grad():
                    boxes = sel.finish_fused(pending, targets)
            return boxes, {"loss_objectness": loss_objectness, "loss_rpn_box_reg": loss_rpn_box_reg}
        else:
            with torch.no_grad():
                boxes = sel(anchors, objectness, rpn_box_regression, targets)
        loss_objectness, loss_rpn_box_reg = self.loss_evaluator(anchors, objectness, rpn_box_regression, targets)
        return boxes, {"loss_objectness": loss_objectness, "loss_rpn_box_reg": loss_rpn_box_reg}

    def _forward_test(self, anchors, objectness, rpn_box_regression):
        boxes = self.box_selector_test(anchors, objectness, rpn_box_regression)
        if cfg.MODEL.RPN_ONLY:
            boxes = [b[b.get_field("objectness").sort(descending=True)[1]] for b in boxes]
        return boxes, {}


def build_rpn(dim_in):
    return RPNModule(dim_in)
