"""CPU restatement of a WHOLE CPM R-CNN iteration: the network arithmetic of oracle/cpu_model.py plus the detection
glue between the networks -- anchors, proposal selection with NMS, matching, sampling, grid targets, the grid decoder,
multi-label NMS and the ISM / RSM re-scoring -- built on the C oracle (oracle/cpm_oracle.c, pinned against the
reference: tests/test_oracle_golden.py) and torch-CPU.

TEST INFRASTRUCTURE ONLY (see cpm_oracle.c): bench.py's `cpu_baseline` legs time it on the GPU box's host cores
((a) forward-only, one image per forward -- BASELINE config #1; (b) forward + backward, bs = 2)
and tests/test_cpu_pipeline.py runs it at toy size.  The product package never imports it.

Every block cites the reference lines it follows.  Random draws (sampler) use numpy's generator: the reference's
torch.randperm stream cannot be reproduced, the sample SIZES follow its rules (pyoracle.balanced_sample_quotas)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import cpu_model as M
from . import pyoracle as O

ANCHOR_SIZES, ANCHOR_STRIDES, ASPECT_RATIOS = (32, 64, 128, 256, 512), (4, 8, 16, 32, 64), (0.5, 1.0, 2.0)
STAGE_IOU, STAGE_RATIO, STAGE_WEIGHT = (0.5, 0.6, 0.7), (1.0, 0.5, 0.25), (1.0, 0.5, 0.25)


def anchors_for(feats):
    """AnchorGenerator.forward, pet/rcnn/modeling/rpn/anchor_generator.py:112-125: one size per level, 3 ratios."""
    out = []
    for f, size, stride in zip(feats, ANCHOR_SIZES, ANCHOR_STRIDES):
        cell = O.cell_anchors(stride, (size,), ASPECT_RATIOS)
        out.append(O.grid_anchors(tuple(f.shape[-2:]), stride, cell))
    return out


def _flat(t, n, c):
    """permute_and_flatten, pet/rcnn/utils/misc.py:6-10: [N, A*C, H, W] -> [N, H*W*A, C]."""
    N, AC, H, W = t.shape
    return t.view(N, AC // c, c, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, c)


def rpn_proposals(lo, br, anchors, img_wh, pre_n, post_n, fpn_post_n, train, gts=None, thr=0.7):
    """RPNPostProcessor, pet/rcnn/modeling/rpn/inference.py:67-172: per level top-k, decode, clip, NMS (the C oracle's
    greedy NMS), post-NMS top-n; cross-level top-k per BATCH in training (:152-163), per image in testing; gts appended
    in training (:42-64).  -> per image (boxes [n,4], objectness [n])."""
    N = lo[0].shape[0]
    w, h = img_wh
    per_img = [([], []) for _ in range(N)]
    for l, (o, b, a) in enumerate(zip(lo, br, anchors)):
        sc = torch.sigmoid(_flat(o.detach(), N, 1).squeeze(2))
        reg = _flat(b.detach(), N, 4)
        k = min(pre_n, sc.shape[1])
        top, idx = sc.topk(k, dim=1)
        for n in range(N):
            boxes = O.box_decode(reg[n, idx[n]].numpy(), a[idx[n].numpy()])
            boxes[:, 0::2] = boxes[:, 0::2].clip(0, w - 1)
            boxes[:, 1::2] = boxes[:, 1::2].clip(0, h - 1)
            s = top[n].numpy()
            keep = O.nms(boxes, s, thr)[:post_n]
            per_img[n][0].append(boxes[keep])
            per_img[n][1].append(s[keep])
    boxes = [np.concatenate(p[0]) for p in per_img]
    obj = [np.concatenate(p[1]) for p in per_img]
    if train:
        allo = np.concatenate(obj)
        k = min(fpn_post_n, allo.size)
        thr_idx = np.argsort(-allo, kind="stable")[:k]
        mask = np.zeros(allo.size, bool)
        mask[thr_idx] = True
        o = 0
        for n in range(N):
            m = mask[o:o + obj[n].size]
            o += obj[n].size
            boxes[n], obj[n] = boxes[n][m], obj[n][m]
            if gts is not None:
                boxes[n] = np.concatenate([boxes[n], gts[n]]).astype(np.float32)
                obj[n] = np.concatenate([obj[n], np.ones(len(gts[n]), np.float32)])
    else:
        for n in range(N):
            order = np.argsort(-obj[n], kind="stable")[:fpn_post_n]
            boxes[n], obj[n] = boxes[n][order], obj[n][order]
    return list(zip(boxes, obj))


def _sample(rng, labels, batch, frac):
    """BalancedPositiveNegativeSampler, balanced_positive_negative_sampler.py:36-62 (sizes per its rules)."""
    pos, neg = np.flatnonzero(labels >= 1), np.flatnonzero(labels == 0)
    npos = min(pos.size, int(batch * frac))
    nneg = min(neg.size, batch - npos)
    return np.sort(np.concatenate([rng.permutation(pos)[:npos], rng.permutation(neg)[:nneg]]))


def rpn_loss(lo, br, anchors, gts, img_wh, rng):
    """RPNLossComputation, pet/rcnn/modeling/rpn/loss.py:53-126: IoU(+1) of every anchor with the gts, Matcher(0.7, 0.3,
    low-quality matches), 256 samples per image at <= 50 % positives, BCE on objectness + smooth-L1 (beta 1/9) on the
    encoded deltas of the positives, both / number of samples."""
    N = lo[0].shape[0]
    obj = torch.cat([_flat(o, N, 1) for o in lo], 1).squeeze(2)
    reg = torch.cat([_flat(b, N, 4) for b in br], 1)
    anc = np.concatenate(anchors)
    w, h = img_wh
    inside = (anc[:, 0] >= 0) & (anc[:, 1] >= 0) & (anc[:, 2] < w) & (anc[:, 3] < h)     # STRADDLE_THRESH = 0
    sel_o, sel_t, pos_r, pos_t = [], [], [], []
    total = 0
    for n in range(N):
        m = O.matcher(O.boxlist_iou(gts[n], anc), 0.7, 0.3, True)
        lab = (m >= 0).astype(np.int64)
        lab[m == -1] = 0
        lab[m == -2] = -1
        lab[~inside] = -1
        idx = _sample(rng, lab, 256, 0.5)
        total += idx.size
        sel_o.append(obj[n, idx])
        sel_t.append(torch.from_numpy((lab[idx] >= 1).astype(np.float32)))
        p = idx[lab[idx] >= 1]
        pos_r.append(reg[n, p])
        pos_t.append(torch.from_numpy(O.box_encode(gts[n][m[p]], anc[p])))
    d = (torch.cat(pos_r) - torch.cat(pos_t)).abs()
    beta = 1.0 / 9
    box = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta).sum() / max(total, 1)
    return F.binary_cross_entropy_with_logits(torch.cat(sel_o), torch.cat(sel_t)), box


def _rois(per_img_boxes):
    return torch.from_numpy(np.concatenate([np.concatenate([np.full((len(b), 1), i, np.float32), b], 1)
                                            for i, b in enumerate(per_img_boxes)]).astype(np.float32))


def _match_labels(boxes, gt, gt_labels, hi, lo_):
    m = O.matcher(O.boxlist_iou(gt, boxes), hi, lo_, False)
    lab = gt_labels[m.clip(min=0)].astype(np.int64)
    lab[m == -1] = 0
    lab[m == -2] = -1
    return m, lab


def train_step(sd, images, gts, gt_labels, rng, layers=(3, 4, 6, 3), backward=True):
    """One training iteration of R-50/101-FPN CPM R-CNN at batch size len(gts) on the CPU
    (tools/rcnn/train_net.py:62-78 -> model_builder.py:71-159 -> grid_cascade_rcnn.py:57-224): forward, the 8 losses,
    backward.  The optimizer step is left out (153 M multiply-adds next to ~3 T)."""
    B, _, H, W = images.shape
    feats = M.fpn(sd, M.backbone(sd, images, layers))
    lo, br = M.rpn_head(sd, feats)
    anchors = anchors_for(feats)
    losses = {}
    losses["loss_objectness"], losses["loss_rpn_box_reg"] = rpn_loss(lo, br, anchors, gts, (W, H), rng)
    props = rpn_proposals(lo, br, anchors, (W, H), 2000, 2000, 2000, True, gts)
    # ---- cls head: 512 samples per image at <= 25 % positives (grid_cascade_rcnn/loss.py:18-110)
    cls_boxes, cls_labels = [], []
    for n, (b, _) in enumerate(props):
        _, lab = _match_labels(b, gts[n], gt_labels[n], 0.5, 0.5)
        idx = _sample(rng, lab, 512, 0.25)
        cls_boxes.append(b[idx])
        cls_labels.append(lab[idx])
    logits = M.cls_head(sd, feats, _rois(cls_boxes))
    losses["loss_classifier"] = F.cross_entropy(logits, torch.from_numpy(np.concatenate(cls_labels)))
    # ---- CMM cascade (grid_cascade_rcnn.py:117-193; loss.py:113-276; inference.py:127-298)
    cur = []
    for n in range(B):
        pos = np.flatnonzero(cls_labels[n] > 0)
        if pos.size > 96:
            pos = rng.permutation(pos)[:96]                               # keep_only_positive_boxes, misc.py:56-80
        cur.append(cls_boxes[n][pos])
    counts = {"cls": int(sum(len(b) for b in cls_boxes))}
    for s in range(3):
        kept, kept_gt, ious = [], [], []
        for n in range(B):
            q = O.boxlist_iou(gts[n], cur[n]) if len(cur[n]) else np.zeros((len(gts[n]), 0), np.float32)
            m = O.matcher(q, STAGE_IOU[s], STAGE_IOU[s], False) if q.shape[1] else np.zeros(0, np.int64)
            pos = m >= 0
            if s != 0:
                cur[n], m = cur[n][pos], m[pos]
                ious.append(q[:, pos].max(0) if pos.any() else np.zeros(0, np.float32))
            else:
                ious.append(q[:, pos].max(0) if pos.any() else np.zeros(0, np.float32))
            kept.append(cur[n])
            kept_gt.append(gts[n][m.clip(min=0)])
        counts["grid_%d" % s] = int(sum(len(b) for b in kept))
        rois = _rois(kept)
        _, heat, iou = M.grid_stage(sd, feats, rois, s, last=(s == 2))
        boxes_all, gt_all = np.concatenate(kept), np.concatenate(kept_gt)
        tgt = torch.from_numpy(O.grid_targets(boxes_all, gt_all, 9, 56, 1, STAGE_RATIO[s]))
        losses["loss_grid_%d" % (s + 1)] = 15 * F.binary_cross_entropy_with_logits(heat, tgt) * STAGE_WEIGHT[s]
        if s == 2:
            # ISM (loss.py:166-176, 271-273; l2_loss.py:4-11): target [1 - iou, iou] of the stage's POSITIVES
            fg = torch.from_numpy(np.concatenate(ious).astype(np.float32))
            t = torch.stack([1 - fg, fg], 1)
            pos_inds = torch.nonzero(t > 0.0).squeeze(1)                  # [P, 2] (row, column) pairs, as l2_loss.py:5
            if pos_inds.shape[0] > 0:
                losses["loss_iou_3"] = (0.5 * (iou[pos_inds] - t[pos_inds]).abs() ** 2 / pos_inds.shape[0]).sum()
            else:
                losses["loss_iou_3"] = (iou * 0.0).sum()
            break
        prob = torch.sigmoid(heat.detach()).numpy()
        o = 0
        for n in range(B):
            k = len(kept[n])
            b, pr = kept[n], prob[o:o + k]
            o += k
            hit = (b[:, None, :] == gts[n][None, :, :]).any(1)
            r = np.where(hit, -1.0, b)
            keep = np.flatnonzero(r.sum(1) > 0)                           # _filter_boxes, inference.py:281-290
            ref = O.grid_decode(b[keep], pr[keep], 9, 56, STAGE_RATIO[s]) if keep.size else np.zeros((0, 4), np.float32)
            cur[n] = np.concatenate([ref, gts[n]]).astype(np.float32)     # add_gt_proposals, :292-298
    # ---- RSM: cls negatives + refined positives -> second cls head (grid_cascade_rcnn.py:195-245)
    rs_boxes, rs_labels = [], []
    for n in range(B):
        b = np.concatenate([cls_boxes[n][cls_labels[n] <= 0], cur[n]])
        _, lab = _match_labels(b, gts[n], gt_labels[n], 0.5, 0.5)
        idx = _sample(rng, lab, 512, 0.25)
        rs_boxes.append(b[idx])
        rs_labels.append(lab[idx])
    counts["rescore"] = int(sum(len(b) for b in rs_boxes))
    logits = M.cls_head(sd, feats, _rois(rs_boxes), "Head_rescore", "Output_rescore")
    losses["loss_rescore"] = F.cross_entropy(logits, torch.from_numpy(np.concatenate(rs_labels)))
    total = sum(losses.values())
    if backward:
        total.backward()
    return {k: float(v.detach()) for k, v in losses.items()}, counts


@torch.no_grad()
def infer_image(sd, image, layers=(3, 4, 6, 3), score_thresh=0.03, nms_thr=0.3):
    """Test-time forward of ONE image (the reference's inference is per image: TEST.IMS_PER_GPU = 1):
    model_builder.py:161-195 -> RPN (1000 pre / post NMS) -> cls head -> CLSPostProcessor (inference.py:59-124: softmax,
    score > 0.03 & label != 0, multi-label NMS 0.3) -> three grid stages refining the kept boxes (:127-298) -> ISM
    (score x IoU logit, :172-182) -> RSM (s^0.8 * p^0.2, :62-76).  -> (boxes [n,4], scores [n], labels [n])."""
    _, _, H, W = image.shape
    feats = M.fpn(sd, M.backbone(sd, image, layers))
    lo, br = M.rpn_head(sd, feats)
    (boxes, _), = rpn_proposals(lo, br, anchors_for(feats), (W, H), 1000, 1000, 1000, False)
    prob = torch.softmax(M.cls_head(sd, feats, _rois([boxes])), -1).numpy()
    K = prob.shape[1]
    cand_boxes = np.repeat(boxes, K, 0)
    cand_boxes[:, 0::2] = cand_boxes[:, 0::2].clip(0, W - 1)
    cand_boxes[:, 1::2] = cand_boxes[:, 1::2].clip(0, H - 1)
    scores, labels = prob.reshape(-1), np.tile(np.arange(K), len(boxes))
    m = (scores > score_thresh) & (labels != 0)
    cb, cs, cl = cand_boxes[m], scores[m], labels[m]
    keep = O.ml_nms(cb, cs, cl, nms_thr)
    b, s, l = cb[keep].astype(np.float32), cs[keep], cl[keep]
    if len(b) == 0:
        return b, s, l
    for st in range(3):
        _, heat, iou = M.grid_stage(sd, feats, _rois([b]), st, last=(st == 2))
        b = O.grid_decode(b, torch.sigmoid(heat).numpy(), 9, 56, STAGE_RATIO[st])
        if st == 2:
            s = s * iou[:, 1].numpy()
    p = torch.softmax(M.cls_head(sd, feats, _rois([b]), "Head_rescore", "Output_rescore"), -1).numpy()
    # inference.py:62-76: `scores ** 0.8` as the reference writes it -- NaN where the ISM-merged score is negative (an
    # untrained ISM branch; torch's pow does the same), no abs()
    with np.errstate(invalid="ignore"):
        s = (s.astype(np.float32) ** np.float32(0.8)) * (p[np.arange(len(b)), l] ** np.float32(0.2))
    return b, s, l
