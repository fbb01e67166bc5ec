"""CPU restatement of the network arithmetic of the CPM R-CNN hot path (torch-CPU fp32 + the C oracle's RoIAlign).

TEST INFRASTRUCTURE ONLY (see cpm_oracle.c): used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
as the checker / the timed CPU baseline.  The product package never imports it.

It is a functional re-statement keyed by the reference's state-dict names; each block cites the reference lines
it follows.  conv / GroupNorm / conv_transpose / linear arithmetic is torch's own (third-party in the reference
too), RoIAlign is oracle/cpm_oracle.c (bit-exact against the reference's ROIAlign_cpu.cpp).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import pyoracle as O


class _RoIAlignCPU(torch.autograd.Function):
    """pet/lib/ops/roi_align.py:14-60 over the C oracle."""

    @staticmethod
    def forward(ctx, x, rois, out, scale, ratio):
        ctx.save_for_backward(rois)
        ctx.cfg = (out, scale, ratio, tuple(x.shape))
        y = O.roi_align_forward(x.detach().numpy(), rois.numpy(), scale, out, out, ratio, False, 0)
        return torch.from_numpy(y)

    @staticmethod
    def backward(ctx, g):
        rois, = ctx.saved_tensors
        out, scale, ratio, (b, c, h, w) = ctx.cfg
        gi = O.roi_align_backward(g.contiguous().numpy(), rois.numpy(), scale, out, out, b, c, h, w, ratio, False, 0)
        return torch.from_numpy(gi), None, None, None, None


def pooler(feats, rois, out, scales, ratio=2):
    """Pooler.forward + LevelMapper, pet/rcnn/utils/poolers.py:30-40,113-132."""
    lv = torch.from_numpy(O.level_map(rois[:, 1:].numpy(), 2, 1 + len(scales)))
    res = torch.zeros((rois.shape[0], feats[0].shape[1], out, out))
    for l, (f, s) in enumerate(zip(feats, scales)):
        idx = torch.nonzero(lv == l).squeeze(1)
        if idx.numel():
            res = res.index_put((idx,), _RoIAlignCPU.apply(f, rois[idx], out, s, ratio))
    return res


def _aff(sd, p, x):
    # AffineChannel2d, pet/lib/ops/affine.py:15-17
    return x * sd[p + ".weight"].view(1, -1, 1, 1) + sd[p + ".bias"].view(1, -1, 1, 1)


def bottleneck(sd, p, x, stride):
    """Bottleneck.forward, pet/models/imagenet/resnet.py:114-136 (stride on the 1x1, STRIDE_3X3=False)."""
    out = F.relu(_aff(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride)))
    out = F.relu(_aff(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1)))
    out = _aff(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
    res = x
    if p + ".downsample.0.weight" in sd:
        res = _aff(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride))
    return F.relu(out + res)


def backbone(sd, x, layers=(3, 4, 6, 3)):
    """ResNet.forward, pet/rcnn/modeling/backbone/ResNet.py:123-148."""
    x = F.relu(_aff(sd, "Conv_Body.bn1", F.conv2d(x, sd["Conv_Body.conv1.weight"], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, n in enumerate(layers):
        for b in range(n):
            x = bottleneck(sd, "Conv_Body.layer%d.%d" % (li + 1, b), x, 2 if (b == 0 and li > 0) else 1)
        outs.append(x)
    return outs


class _DeformConvCPU(torch.autograd.Function):
    """pet/lib/ops/deform_conv.py:13-145 over the C oracle (orc_deform_conv)."""

    @staticmethod
    def forward(ctx, x, offset, w, stride, pad, dil, groups, dg):
        ctx.cfg = (stride, pad, dil, groups, dg)
        ctx.save_for_backward(x, offset, w)
        return torch.from_numpy(O.deform_conv(x.detach().numpy(), offset.detach().numpy(), w.detach().numpy(),
                                              stride, pad, dil, groups, dg))

    @staticmethod
    def backward(ctx, g):
        x, offset, w = ctx.saved_tensors
        stride, pad, dil, groups, dg = ctx.cfg
        _, dx, doff, dw = O.deform_conv(x.detach().numpy(), offset.detach().numpy(), w.detach().numpy(), stride,
                                        pad, dil, groups, dg, dy=g.contiguous().numpy())
        return torch.from_numpy(dx), torch.from_numpy(doff), torch.from_numpy(dw), None, None, None, None, None


def resnext_bottleneck(sd, p, x, stride, groups):
    """Bottleneck.forward, pet/models/imagenet/resnext.py:61-83: stride on the grouped 3x3; that conv is a
    DeformConvPack (deform_conv.py:500-512) when the block has a conv_offset child."""
    out = F.relu(_aff(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    if p + ".conv2.conv_offset.weight" in sd:
        off = F.conv2d(out, sd[p + ".conv2.conv_offset.weight"], sd[p + ".conv2.conv_offset.bias"], stride, 1)
        out = _DeformConvCPU.apply(out, off, sd[p + ".conv2.weight"], stride, 1, 1, groups, 1)
    else:
        out = F.conv2d(out, sd[p + ".conv2.weight"], None, stride, 1, 1, groups)
    out = F.relu(_aff(sd, p + ".bn2", out))
    out = _aff(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
    res = x
    if p + ".downsample.0.weight" in sd:
        res = _aff(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride))
    return F.relu(out + res)


def resnext_backbone(sd, x, layers=(3, 4, 23, 3), groups=64):
    """ResNeXt.forward, pet/rcnn/modeling/backbone/ResNeXt.py:107-132."""
    x = F.relu(_aff(sd, "Conv_Body.bn1", F.conv2d(x, sd["Conv_Body.conv1.weight"], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, n in enumerate(layers):
        for b in range(n):
            x = resnext_bottleneck(sd, "Conv_Body.layer%d.%d" % (li + 1, b), x, 2 if (b == 0 and li > 0) else 1,
                                   groups)
        outs.append(x)
    return outs


def fpn(sd, c):
    """fpn.forward, pet/rcnn/modeling/fpn/FPN.py:96-121."""
    P = "Conv_Body_FPN."
    px = F.conv2d(c[-1], sd[P + "p5_in.weight"], sd[P + "p5_in.bias"])
    outs = [F.conv2d(px, sd[P + "p5_out.weight"], sd[P + "p5_out.bias"], 1, 1)]
    for i in range(3):
        lat = F.conv2d(c[-i - 2], sd[P + "fpn_in.%d.weight" % i], sd[P + "fpn_in.%d.bias" % i])
        px = lat + F.interpolate(px, scale_factor=2, mode="nearest")
        outs.insert(0, F.conv2d(px, sd[P + "fpn_out.%d.weight" % i], sd[P + "fpn_out.%d.bias" % i], 1, 1))
    outs.append(F.max_pool2d(outs[-1], 1, 2, 0))
    return outs


def rpn_head(sd, feats):
    """RPNHead.forward, pet/rcnn/modeling/rpn/rpn.py:34-41."""
    lo, br = [], []
    for f in feats:
        t = F.relu(F.conv2d(f, sd["RPN.head.conv.weight"], sd["RPN.head.conv.bias"], 1, 1))
        lo.append(F.conv2d(t, sd["RPN.head.cls_logits.weight"], sd["RPN.head.cls_logits.bias"]))
        br.append(F.conv2d(t, sd["RPN.head.bbox_pred.weight"], sd["RPN.head.bbox_pred.bias"]))
    return lo, br


SCALES = (1 / 4., 1 / 8., 1 / 16., 1 / 32.)


def cls_head(sd, feats, rois, head="Head_cls", out="Output_cls"):
    """roi_cls_head + Cls_output, grid_rcnn/heads/cls_heads.py:40-48, outputs.py:98-104."""
    G = "Grid_Cascade_RCNN."
    x = pooler(feats[:4], rois, 7, SCALES).flatten(1)
    x = F.relu(F.linear(x, sd[G + head + ".fc6.weight"], sd[G + head + ".fc6.bias"]))
    x = F.relu(F.linear(x, sd[G + head + ".fc7.weight"], sd[G + head + ".fc7.bias"]))
    return F.linear(x, sd[G + out + ".cls_score.weight"], sd[G + out + ".cls_score.bias"])


def grid_stage(sd, feats, rois, stage, points=9, last=False):
    """roi_grid_head + Grid_output, grid_heads.py:131-160, outputs.py:49-83."""
    H, Oo = "Grid_Cascade_RCNN.Head_grid_%d." % stage, "Grid_Cascade_RCNN.Output_grid_%d." % stage
    x = pooler(feats[:4], rois, 14, SCALES)
    for j in range(8):
        x = F.conv2d(x, sd[H + "convs.%d.0.weight" % j], sd[H + "convs.%d.0.bias" % j], 2 if j == 0 else 1, 1)
        x = F.relu(F.group_norm(x, 4 * points, sd[H + "convs.%d.1.weight" % j], sd[H + "convs.%d.1.bias" % j], 1e-5))
    y = F.conv_transpose2d(x, sd[Oo + "deconv_1.weight"], sd[Oo + "deconv_1.bias"], 2, 1, groups=points)
    y = F.relu(F.group_norm(y, points, sd[Oo + "norm1.weight"], sd[Oo + "norm1.bias"], 1e-5))
    heat = F.conv_transpose2d(y, sd[Oo + "deconv_2.weight"], sd[Oo + "deconv_2.bias"], 2, 1, groups=points)
    iou = None
    if last:
        t = F.relu(F.linear(x.flatten(1), sd[Oo + "iou_fc1.weight"], sd[Oo + "iou_fc1.bias"]))
        t = F.relu(F.linear(t, sd[Oo + "iou_fc2.weight"], sd[Oo + "iou_fc2.bias"]))
        iou = F.linear(t, sd[Oo + "iou_pred.weight"], sd[Oo + "iou_pred.bias"])
    return x, heat, iou


def train_step_compute(sd, image, rois_cls, rois_grid, rois_rescore, layers=(3, 4, 6, 3)):
    """The conv / FC / RoIAlign work of one training iteration (forward + backward) at given RoI sets -- the
    CPU baseline's timed body.  Losses are stand-in sums of squares: the baseline measures the tensor work, the
    matching / sampling / NMS glue (negligible next to ~1 TMAC of convs) is left out and said so in bench.py."""
    c = backbone(sd, image, layers)
    p = fpn(sd, c)
    lo, br = rpn_head(sd, p)
    loss = sum((a ** 2).mean() for a in lo) + sum((a ** 2).mean() for a in br)
    loss = loss + (cls_head(sd, p, rois_cls) ** 2).mean()
    for s, r in enumerate(rois_grid):
        _, heat, iou = grid_stage(sd, p, r, s, last=(s == len(rois_grid) - 1))
        loss = loss + (heat ** 2).mean() + ((iou ** 2).mean() if iou is not None else 0)
    loss = loss + (cls_head(sd, p, rois_rescore, "Head_rescore", "Output_rescore") ** 2).mean()
    loss.backward()
    return float(loss.detach())
