"""Box (de)coding for the RPN (counterpart of pet/rcnn/utils/box_coder.py:6-94).  Pixel-inclusive widths
(+1), dw/dh clipped at log(1000/16) before exp, x2/y2 get the "-1" back (box_coder.py:88-92)."""
import math

import torch


class BoxCoder(object):
    def __init__(self, weights, bbox_xform_clip=math.log(1000. / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def encode(self, reference_boxes, proposals):
        pw = proposals[:, 2] - proposals[:, 0] + 1
        ph = proposals[:, 3] - proposals[:, 1] + 1
        pcx = proposals[:, 0] + 0.5 * pw
        pcy = proposals[:, 1] + 0.5 * ph
        gw = reference_boxes[:, 2] - reference_boxes[:, 0] + 1
        gh = reference_boxes[:, 3] - reference_boxes[:, 1] + 1
        gcx = reference_boxes[:, 0] + 0.5 * gw
        gcy = reference_boxes[:, 1] + 0.5 * gh
        wx, wy, ww, wh = self.weights
        return torch.stack((wx * (gcx - pcx) / pw, wy * (gcy - pcy) / ph, ww * torch.log(gw / pw),
                            wh * torch.log(gh / ph)), dim=1)

    def decode(self, rel_codes, boxes):
        boxes = boxes.to(rel_codes.dtype)
        w = boxes[:, 2] - boxes[:, 0] + 1
        h = boxes[:, 3] - boxes[:, 1] + 1
        cx = boxes[:, 0] + 0.5 * w
        cy = boxes[:, 1] + 0.5 * h
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = torch.clamp(rel_codes[:, 2::4] / ww, max=self.bbox_xform_clip)
        dh = torch.clamp(rel_codes[:, 3::4] / wh, max=self.bbox_xform_clip)
        pcx = dx * w[:, None] + cx[:, None]
        pcy = dy * h[:, None] + cy[:, None]
        pw = torch.exp(dw) * w[:, None]
        ph = torch.exp(dh) * h[:, None]
        out = torch.zeros_like(rel_codes)
        out[:, 0::4] = pcx - 0.5 * pw
        out[:, 1::4] = pcy - 0.5 * ph
        out[:, 2::4] = pcx + 0.5 * pw - 1
        out[:, 3::4] = pcy + 0.5 * ph - 1
        return out
