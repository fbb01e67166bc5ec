"""RPN proposal selection (counterpart of pet/rcnn/modeling/rpn/inference.py:12-196).

Same pipeline -- per level: sigmoid, top-k, BoxCoder.decode, clip, min-size filter, NMS, post-NMS top-n; then the
cross-level top-k (per batch in training, per image in testing) and the GT append -- but the reference's
N_images x N_levels separate NMS calls (each with a device->host mask copy) become ONE batched device NMS
(cpm_nms_batched) over all (image, level) segments."""
import os

import numpy as np
import torch

import pet.lib.ops as ops
from pet.lib.ops import nms_segments
from pet.lib.ops import roi_lists as RL
from pet.rcnn.core.config import cfg
from pet.rcnn.utils.box_coder import BoxCoder
from pet.rcnn.utils.misc import permute_and_flatten
from pet.utils.data.structures.bounding_box import BoxList
from pet.utils.data.structures.boxlist_ops import cat_boxlist


class RPNPostProcessor(torch.nn.Module):
    def __init__(self, pre_nms_top_n, post_nms_top_n, nms_thresh, min_size, box_coder=None, fpn_post_nms_top_n=None,
                 fpn_post_nms_per_batch=True):
        super().__init__()
        self.pre_nms_top_n, self.post_nms_top_n = pre_nms_top_n, post_nms_top_n
        self.nms_thresh, self.min_size = nms_thresh, min_size
        self.box_coder = box_coder if box_coder is not None else BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.fpn_post_nms_top_n = post_nms_top_n if fpn_post_nms_top_n is None else fpn_post_nms_top_n
        self.fpn_post_nms_per_batch = fpn_post_nms_per_batch
        # CPM_FUSED_GLUE=0 runs the per-level / per-image tensor-op formulation below (the in-tree cross-check)
        self.fused_glue = os.environ.get("CPM_FUSED_GLUE", "1") != "0"
        # CPM_DEVICE_LISTS=0: proposals come back as per-image BoxLists built from host index lists (two round trips)
        self.device_lists = self.fused_glue and os.environ.get("CPM_DEVICE_LISTS", "1") != "0"

    def add_gt_proposals(self, proposals, targets):
        device = proposals[0].bbox.device
        out = []
        for p, t in zip(proposals, targets):
            gt = t.copy_with_fields([])
            gt.add_field("objectness", torch.ones(len(gt), device=device))
            out.append(cat_boxlist((p, gt)))
        return out

    def _candidates(self, anchors, objectness, box_regression):
        """One level: [N, k] scores and [N, k, 4] decoded + clipped boxes (inference.py:67-99)."""
        N, A, H, W = objectness.shape
        scores = permute_and_flatten(objectness, N, A, 1, H, W).view(N, -1).sigmoid()
        reg = permute_and_flatten(box_regression, N, A, 4, H, W)
        k = min(self.pre_nms_top_n, A * H * W)
        scores, idx = scores.topk(k, dim=1, sorted=True)
        bidx = torch.arange(N, device=scores.device)[:, None]
        reg = reg[bidx, idx]
        anc = torch.cat([a.bbox for a in anchors], dim=0).reshape(N, -1, 4)[bidx, idx]
        boxes = self.box_coder.decode(reg.reshape(-1, 4), anc.reshape(-1, 4)).view(N, k, 4)
        for n, a in enumerate(anchors):                       # clip_to_image(remove_empty=False)
            w, h = a.size
            boxes[n, :, 0].clamp_(min=0, max=w - 1)
            boxes[n, :, 1].clamp_(min=0, max=h - 1)
            boxes[n, :, 2].clamp_(min=0, max=w - 1)
            boxes[n, :, 3].clamp_(min=0, max=h - 1)
        return scores, boxes

    def start_fused(self, anchors, objectness, box_regression, read_counts=True):
        """Same selection as forward() below with the per-level decode in one kernel (cpm_rpn_decode), every
        post-NMS gather done once for the whole batch from host-built index lists, and two host round trips in all
        (NMS counts, cross-level top-k mask) instead of one per image and boolean index."""
        num_levels, N = len(objectness), objectness[0].shape[0]
        dev = objectness[0].device
        sizes = [per_img[0].size for per_img in anchors]
        lvl_k, lvl_n, offsets, owner = [], [], [0], []
        for o in objectness:
            _, A, H, W = o.shape
            lvl_n.append(A * H * W)
            lvl_k.append(min(self.pre_nms_top_n, A * H * W))
            for n in range(N):                                # segments: level-major, then image
                offsets.append(offsets[-1] + lvl_k[-1])
                owner.append(n)
        lvl_off = offsets[0::N]
        regs = [permute_and_flatten(b, N, b.shape[1] // 4, 4, b.shape[2], b.shape[3]) for b in box_regression]
        dense = all(o.permute(0, 2, 3, 1).is_contiguous() for o in objectness) and all(r.is_contiguous() for r in regs)
        if max(lvl_k) <= ops.detect_glue.TOPK_MAX and num_levels <= 8 and dense:
            # three launches for all levels and images: sigmoid, row-wise top-k, decode -- the last two writing
            # straight into the segment layout the batched NMS reads (no per-level tensors, no cat)
            lvl_scores = [s.view(N, n) for s, n in zip(ops.sigmoid_multi(objectness), lvl_n)]
            all_scores = torch.empty((offsets[-1],), dtype=torch.float32, device=dev)
            all_idx = torch.empty((offsets[-1],), dtype=torch.int64, device=dev)
            all_boxes = torch.empty((offsets[-1], 4), dtype=torch.float32, device=dev)
            out = [(all_scores[lvl_off[l]:lvl_off[l + 1]].view(N, lvl_k[l]), all_idx[lvl_off[l]:lvl_off[l + 1]].view(N, lvl_k[l]))
                   for l in range(num_levels)]
            ops.topk_rows_multi(lvl_scores, lvl_k, out=out)
            ops.rpn_decode_multi(regs, [o[1] for o in out], [a.bbox for a in anchors[0]], lvl_off[:-1], all_boxes,
                                 self.box_coder.weights, self.box_coder.bbox_xform_clip, sizes)
        else:
            seg_boxes, seg_scores = [], []
            for lvl, o in enumerate(objectness):
                scores, idx = permute_and_flatten(o, N, o.shape[1], 1, o.shape[2], o.shape[3]).view(N, -1).sigmoid() \
                    .topk(lvl_k[lvl], dim=1, sorted=True)
                boxes = ops.rpn_decode(regs[lvl], idx, anchors[0][lvl].bbox, self.box_coder.weights,
                                       self.box_coder.bbox_xform_clip, sizes)
                seg_boxes.append(boxes.view(N * lvl_k[lvl], 4))
                seg_scores.append(scores.reshape(N * lvl_k[lvl]))
            all_boxes, all_scores = torch.cat(seg_boxes, 0), torch.cat(seg_scores, 0)
        # (every segment is a row of a sorted top-k: its scores descend, the NMS's stable sort would change nothing)
        keep, counts = nms_segments(all_boxes, all_scores, None, offsets, self.nms_thresh, 0, presorted=True)
        # host round trip 1, split in two: the kept counts travel to pinned memory behind the NMS kernels and an
        # event marks the copy; the caller may queue unrelated device work (the RPN loss) before finish() waits for
        # that event only -- the device then stays busy while the host builds the index lists below
        counts_h = ev = None
        if read_counts:
            counts_h = torch.empty(counts.shape, dtype=counts.dtype).pin_memory()
            counts_h.copy_(counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return dict(num_levels=num_levels, N=N, dev=dev, sizes=sizes, offsets=offsets, owner=owner, all_boxes=all_boxes,
                    all_scores=all_scores, keep=keep, counts=counts, counts_h=counts_h, event=ev)

    def can_keep_on_device(self, objectness, targets):
        """training selection (one top-k over the batch, gts appended) as ONE launch with the result left on the
        device as a packed list (pet/lib/ops/roi_lists.py) -- no host round trip in the proposal stage at all"""
        if not (self.device_lists and self.training and targets is not None and self.can_fuse(objectness)
                and (len(objectness) == 1 or self.fpn_post_nms_per_batch)
                and len(objectness) * objectness[0].shape[0] <= 512):
            return False
        # the packed list's capacity bounds every image's rows: it must fit the RoI heads' sampling kernel
        # (cpm_roi_sample: 4096 rows per image); a non-FPN RPN (no batch-wide top-k) or a huge FPN_POST_NMS_TOP_N
        # takes the BoxList path instead
        n_img, n_gt = objectness[0].shape[0], sum(len(t) for t in targets)
        cap = (self.fpn_post_nms_top_n if len(objectness) > 1 else
               n_img * (self.post_nms_top_n if self.post_nms_top_n > 0 else self.pre_nms_top_n)) + n_gt
        return cap <= RL.roi_sample_max_rows()

    def finish_device(self, st, targets):
        gt_all, _, gt_off, off_h = RL.gt_pack(targets)
        k = self.fpn_post_nms_top_n if st["num_levels"] > 1 else (1 << 30)
        return RL.proposals_finalize(st["all_boxes"], st["all_scores"], st["keep"], st["counts"], st["offsets"],
                                     st["N"], st["num_levels"], self.post_nms_top_n, k, gt_all, gt_off, off_h[-1],
                                     st["sizes"])

    def finish_fused(self, st, targets=None):
        num_levels, N, dev, sizes, offsets, owner = (st[k] for k in ("num_levels", "N", "dev", "sizes", "offsets", "owner"))
        all_boxes, all_scores, keep = st["all_boxes"], st["all_scores"], st["keep"]
        st["event"].synchronize()
        counts = st["counts_h"].tolist()
        # rows of `keep` to read, image-major then level-major (= cat_boxlist of the per-level lists), and the
        # segment base to add to the segment-relative indices stored there
        pos, base, per_image = [], [], [0] * N
        for n in range(N):
            for s_, own in enumerate(owner):
                if own != n:
                    continue
                c = counts[s_] if self.post_nms_top_n <= 0 else min(counts[s_], self.post_nms_top_n)
                pos.append(np.arange(offsets[s_], offsets[s_] + c))
                base.append(np.full(c, offsets[s_]))
                per_image[n] += c
        pb = torch.from_numpy(np.stack([np.concatenate(pos), np.concatenate(base)])).pin_memory().to(dev, non_blocking=True)
        sel = keep[pb[0]] + pb[1]                              # absolute rows of all_boxes / all_scores
        obj = all_scores[sel]
        if num_levels > 1:
            if self.training and self.fpn_post_nms_per_batch:
                k2 = min(self.fpn_post_nms_top_n, obj.numel())
                _, inds = torch.topk(obj, k2, dim=0, sorted=True)
                mask = torch.zeros(obj.numel(), dtype=torch.bool, device=dev)
                mask[inds] = True
                mask_h = mask.cpu().numpy()                    # host round trip 2
                chosen, o_ = [], 0
                for n in range(N):
                    idx_n = np.flatnonzero(mask_h[o_:o_ + per_image[n]]) + o_
                    o_ += per_image[n]
                    per_image[n] = len(idx_n)
                    chosen.append(idx_n)
                ch = torch.from_numpy(np.concatenate(chosen)).pin_memory().to(dev, non_blocking=True)
            else:
                chosen, o_ = [], 0
                for n in range(N):
                    seg = obj[o_:o_ + per_image[n]]
                    _, inds = torch.topk(seg, min(self.fpn_post_nms_top_n, seg.numel()), dim=0, sorted=True)
                    chosen.append(inds + o_)
                    o_ += per_image[n]
                    per_image[n] = inds.numel()
                ch = torch.cat(chosen)
            sel, obj = sel[ch], obj[ch]
        boxes = all_boxes[sel]
        if self.training and targets is not None:              # add_gt_proposals: one cat per field for the batch
            bparts, oparts, o_ = [], [], 0
            for n in range(N):
                bparts += [boxes[o_:o_ + per_image[n]], targets[n].bbox]
                oparts += [obj[o_:o_ + per_image[n]], torch.ones(len(targets[n]), device=dev)]
                o_ += per_image[n]
                per_image[n] += len(targets[n])
            boxes, obj = torch.cat(bparts, 0), torch.cat(oparts, 0)
        out, o_ = [], 0
        for n in range(N):
            bl = BoxList(boxes[o_:o_ + per_image[n]], sizes[n], mode="xyxy")
            bl.add_field("objectness", obj[o_:o_ + per_image[n]])
            out.append(bl)
            o_ += per_image[n]
        return out

    def can_fuse(self, objectness):
        return self.fused_glue and self.min_size <= 0 and objectness[0].is_cuda and objectness[0].shape[0] <= 64

    def forward(self, anchors, objectness, box_regression, targets=None):
        if self.can_fuse(objectness):
            return self.finish_fused(self.start_fused(anchors, objectness, box_regression), targets)
        num_levels, N = len(objectness), objectness[0].shape[0]
        per_level_anchors = list(zip(*anchors))
        seg_boxes, seg_scores, offsets, owner = [], [], [0], []
        for lvl, (a, o, b) in enumerate(zip(per_level_anchors, objectness, box_regression)):
            scores, boxes = self._candidates(a, o, b)
            for n in range(N):
                sb, ss = boxes[n], scores[n]
                if self.min_size > 0:       # remove_small_boxes; with MIN_SIZE = 0 every decoded box passes
                    keep = ((sb[:, 2] - sb[:, 0] + 1 >= self.min_size) & (sb[:, 3] - sb[:, 1] + 1 >= self.min_size))
                    sb, ss = sb[keep], ss[keep]
                seg_boxes.append(sb)
                seg_scores.append(ss)
                offsets.append(offsets[-1] + sb.shape[0])
                owner.append(n)
        all_boxes, all_scores = torch.cat(seg_boxes, 0).contiguous(), torch.cat(seg_scores, 0).contiguous()
        keep, counts = nms_segments(all_boxes, all_scores, None, offsets, self.nms_thresh, 0)
        counts = counts.tolist()                               # the one host sync of the proposal stage
        per_image = [[] for _ in range(N)]
        for s, n in enumerate(owner):
            c = counts[s] if self.post_nms_top_n <= 0 else min(counts[s], self.post_nms_top_n)
            sel = keep[offsets[s]: offsets[s] + c] + offsets[s]
            bl = BoxList(all_boxes[sel], anchors[n][0].size, mode="xyxy")
            bl.add_field("objectness", all_scores[sel])
            per_image[n].append(bl)
        boxlists = [cat_boxlist(b) for b in per_image]
        if num_levels > 1:
            boxlists = self.select_over_all_levels(boxlists)
        if self.training and targets is not None:
            boxlists = self.add_gt_proposals(boxlists, targets)
        return boxlists

    def select_over_all_levels(self, boxlists):
        if self.training and self.fpn_post_nms_per_batch:
            obj = torch.cat([b.get_field("objectness") for b in boxlists], dim=0)
            sizes = [len(b) for b in boxlists]
            k = min(self.fpn_post_nms_top_n, len(obj))
            _, inds = torch.topk(obj, k, dim=0, sorted=True)
            mask = torch.zeros_like(obj, dtype=torch.bool)
            mask[inds] = True
            return [b[m] for b, m in zip(boxlists, mask.split(sizes))]
        out = []
        for b in boxlists:
            obj = b.get_field("objectness")
            _, inds = torch.topk(obj, min(self.fpn_post_nms_top_n, len(obj)), dim=0, sorted=True)
            out.append(b[inds])
        return out


def make_rpn_postprocessor(rpn_box_coder, is_train):
    R = cfg.RPN
    return RPNPostProcessor(
        pre_nms_top_n=R.PRE_NMS_TOP_N_TRAIN if is_train else R.PRE_NMS_TOP_N_TEST,
        post_nms_top_n=R.POST_NMS_TOP_N_TRAIN if is_train else R.POST_NMS_TOP_N_TEST,
        nms_thresh=R.NMS_THRESH, min_size=R.MIN_SIZE, box_coder=rpn_box_coder,
        fpn_post_nms_top_n=R.FPN_POST_NMS_TOP_N_TRAIN if is_train else R.FPN_POST_NMS_TOP_N_TEST,
        fpn_post_nms_per_batch=R.FPN_POST_NMS_PER_BATCH)
