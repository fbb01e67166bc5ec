"""CPM head = CMM cascade + ISM + RSM (counterpart of
pet/rcnn/modeling/grid_cascade_rcnn/grid_cascade_rcnn.py:15-309).

Training: sample 512 RoIs/img -> cls head -> CE; keep <= 96 positives/img -> for each of the N stages: match at
the stage's IoU, pool 14x14, 8 convs, 2 deconvs -> BCE against rasterised point targets (last stage: ISM l2
loss), decode refined boxes (no_grad) and append gts for the next stage; RSM: cls negatives + refined positives
-> second cls head -> CE.  Testing: cls head -> ml_nms -> the stages refine the kept detections -> ISM / RSM
re-scoring."""
import os

import numpy as np
import torch
from torch import nn

import pet.lib.ops as ops
from pet.lib.ops import conv as conv_ops
from pet.lib.ops import roi_lists as RL
from pet.rcnn.core.config import cfg
from pet.rcnn.modeling import registry
from pet.rcnn.modeling.grid_cascade_rcnn.inference import post_processor
from pet.rcnn.modeling.grid_cascade_rcnn.loss import loss_evaluator
from pet.rcnn.modeling.grid_rcnn import heads, outputs  # noqa: F401  (populate the registries)
from pet.rcnn.utils.misc import keep_only_positive_boxes
from pet.utils.data.structures.bounding_box import BoxList
from pet.utils.data.structures.boxlist_ops import cat_boxlist


def _to_device_async(values, dtype, device):
    """small host list -> device through pinned memory (an ordinary pageable copy would wait for the stream)"""
    return torch.tensor(values, dtype=dtype).pin_memory().to(device, non_blocking=True)


class GridCascadeRCNN(nn.Module):
    def __init__(self, dim_in, spatial_scale):
        super().__init__()
        G, M = cfg.GRID_RCNN, cfg.GRID_RCNN.CASCADE_MAPPING_OPTION
        if G.ENHANCE_FEATURES or G.EXTEND_ROI or G.RANDOM_JITTER:
            raise ValueError("ENHANCE_FEATURES / EXTEND_ROI / RANDOM_JITTER are outside the hot path")
        self.Head_cls = registry.ROI_CLS_HEADS[G.ROI_CLS_HEAD](dim_in, spatial_scale)
        self.Output_cls = registry.ROI_CLS_OUTPUTS[G.ROI_CLS_OUTPUT](self.Head_cls.dim_out)
        self.cls_post_processor = post_processor(type="cls")
        self.cls_loss_evaluator = loss_evaluator(type="cls")
        self.max_sample_num_grid = G.MAX_SAMPLE_NUM_GRID
        self.test_ensemble, self.stage_num, self.test_stage = M.TEST_ENSEMBLE, M.STAGE_NUM, M.TEST_STAGE
        self.stage_loss_weight = M.STAGE_WEIGHTS
        head_grid = registry.ROI_GRID_HEADS[G.ROI_GRID_HEAD]
        output_grid = registry.ROI_GRID_OUTPUTS[G.ROI_GRID_OUTPUT]
        self.grid_loss_evaluators, self.grid_post_processors = [], []
        for s in range(self.stage_num):
            setattr(self, "Head_grid_%d" % s, head_grid(dim_in, spatial_scale, s))
            setattr(self, "Output_grid_%d" % s, output_grid(getattr(self, "Head_grid_%d" % s).dim_out, s))
            self.grid_loss_evaluators.append(loss_evaluator(stage=s, type="grid"))
            self.grid_post_processors.append(post_processor(stage=s, type="grid"))
        if G.RESCORE_ON:
            self.Head_rescore = registry.ROI_CLS_HEADS[G.ROI_CLS_HEAD](dim_in, spatial_scale)
            self.Output_rescore = registry.ROI_CLS_OUTPUTS[G.ROI_CLS_OUTPUT](self.Head_rescore.dim_out)
            self.rescore_loss_evaluator = loss_evaluator(type="cls")
        self._last_counts, self._pending = {}, None
        # CPM_FUSED_GLUE=0 runs the per-image formulation (kept as the in-tree cross-check of the fused kernels)
        self.fused_glue = os.environ.get("CPM_FUSED_GLUE", "1") != "0"
        # proposals arrive as a packed device list and every RoI set of the step is built on the device
        # (_forward_train_lists); CPM_DEVICE_LISTS=0 keeps the BoxList formulation with its host-built index lists
        self.takes_device_lists = (self.fused_glue and os.environ.get("CPM_DEVICE_LISTS", "1") != "0"
                                   and not (G.FUSED_ON or G.BETTER_ROI or G.TARGET_REFINE or G.ACROSS_SAMPLE
                                            or M.RESIZE_ROI or G.RESCORE_OPTION.KEEP_RATIO))
        self._count_reads = [RL.Counts() for _ in range(M.STAGE_NUM + 2)]

    def forward(self, features, proposals, targets=None):
        if self.training and isinstance(proposals, RL.RoIList):
            return self._forward_train_lists(features, proposals, targets)
        if self.training:
            loss = {}
            proposals, loss_cls = self._forward_train_cls(features, proposals, targets)
            x, result, loss_grid = self._forward_train_cascade(features, proposals, targets)
            if cfg.GRID_RCNN.RESCORE_ON:
                result, loss_rescore = self._forward_train_rescore(features, proposals, result, targets)
                loss.update(loss_rescore)
            loss.update(loss_cls)
            loss.update(loss_grid)
            return x, result, loss
        proposals = self._forward_test_cls(features, proposals)
        if len(proposals[0]) == 0:
            return features, proposals, {}
        x, result = self._forward_test_cascade(features, proposals)
        if cfg.GRID_RCNN.RESCORE_ON:
            result = self._forward_test_rescore(features, result)
        return x, result, {}

    # ---- training ----------------------------------------------------------------------------------------
    def _forward_train_cls(self, features, proposals, targets):
        with torch.no_grad():
            proposals = self.cls_loss_evaluator.subsample(proposals, targets)
        self.last_counts["cls"] = sum(len(p) for p in proposals)
        logits = self.Output_cls(self.Head_cls(features, proposals))
        return proposals, dict(loss_classifier=self.cls_loss_evaluator([logits]))

    def _forward_train_cascade(self, features, proposals, targets):
        G = cfg.GRID_RCNN
        if self.fused_glue and not (G.FUSED_ON or G.BETTER_ROI or G.TARGET_REFINE or G.ACROSS_SAMPLE
                                    or G.CASCADE_MAPPING_OPTION.RESIZE_ROI):
            return self._forward_train_cascade_fused(features, proposals, targets)
        losses, x = {}, None
        for s in range(self.stage_num):
            ev = self.grid_loss_evaluators[s]
            if s == 0:
                proposals = keep_only_positive_boxes(proposals, roi_batch_size=self.max_sample_num_grid,
                                                     across_sample=G.ACROSS_SAMPLE)
            with torch.no_grad():
                proposals = ev.subsample(proposals, targets)
            self.last_counts["grid_%d" % s] = sum(len(p) for p in proposals)
            x, _ = getattr(self, "Head_grid_%d" % s)(features, proposals)
            grid_logits, iou_logits = getattr(self, "Output_grid_%d" % s)(x, None)
            loss_grid, loss_iou = ev(proposals, grid_logits, iou_logits, targets)
            loss_grid = loss_grid * self.stage_loss_weight[s]
            if G.IOU_HELPER and s == self.stage_num - 1:
                losses["loss_iou_%d" % (s + 1)] = loss_iou * G.IOU_LOSS_WEIGHT
            losses["loss_grid_%d" % (s + 1)] = loss_grid
            if s < self.stage_num - 1:
                with torch.no_grad():
                    proposals = self.grid_post_processors[s](grid_logits, proposals, targets=targets, is_train=True)
        return x, proposals, losses

    def _forward_train_cascade_fused(self, features, proposals, targets):
        """Same computation as the loop above (grid_cascade_rcnn.py:117-160 in the reference), restated over the
        whole batch: per stage ONE matching launch (cpm_match_rois), ONE loss launch with the targets rasterised on
        the fly (cpm_grid_bce_loss) and ONE decode+filter launch (cpm_grid_decode), and one host round trip per
        stage transition (the survivors' mask) instead of one per image, field and boolean index.  The RoIs of all
        images travel concatenated; per-image BoxLists are rebuilt (as views) for the heads and at the end."""
        G, M = cfg.GRID_RCNN, cfg.GRID_RCNN.CASCADE_MAPPING_OPTION
        dev = features[0].device
        n_img = len(proposals)
        proposals = keep_only_positive_boxes(proposals, roi_batch_size=self.max_sample_num_grid, across_sample=False)
        sizes = [p.size for p in proposals]
        gt_counts = [len(t) for t in targets]
        gt_off_h = np.concatenate([[0], np.cumsum(gt_counts)]).astype(np.int64)
        n_gt = int(gt_off_h[-1])
        gt_all = torch.cat([t.bbox for t in targets], dim=0)
        gt_off = _to_device_async(gt_off_h.tolist(), torch.int32, dev)
        counts = [len(p) for p in proposals]
        rois = torch.cat([p.bbox for p in proposals], dim=0)
        # per-RoI attributes that only ride along (the RSM stage reads them from the returned BoxLists)
        labels = torch.cat([p.get_field("labels") for p in proposals] + [t.get_field("labels") for t in targets])
        obj = torch.cat([p.get_field("objectness") for p in proposals] + [torch.ones_like(gt_all[:, 0])])
        src = torch.arange(sum(counts), device=dev)        # row of (labels, obj) each current RoI came from
        gt_src = torch.arange(sum(counts), sum(counts) + n_gt, device=dev)

        def roi_img_and_base(cnts):
            """image index and first-gt row of every RoI, built on the host (tiny) and copied asynchronously"""
            im = np.repeat(np.arange(n_img), cnts)
            return (_to_device_async(im.tolist(), torch.int32, dev),
                    _to_device_async(gt_off_h[im].tolist(), torch.int64, dev))

        img, base = roi_img_and_base(counts)
        with torch.no_grad():
            matched, max_iou = ops.match_rois(rois, img, gt_all, gt_off, M.FG_IOU_THRESHOLD[0], M.BG_IOU_THRESHOLD[0])
            gt_boxes = gt_all[matched.clamp(min=0) + base]
        losses, x = {}, None
        for s in range(self.stage_num):
            ev = self.grid_loss_evaluators[s]
            last = s == self.stage_num - 1
            self.last_counts["grid_%d" % s] = sum(counts)
            boxlists = [BoxList(b, sz) for b, sz in zip(rois.split(counts), sizes)]
            x, _ = getattr(self, "Head_grid_%d" % s)(features, boxlists)
            grid_logits, iou_logits = getattr(self, "Output_grid_%d" % s)(x, None)
            logits = grid_logits["unfused"]
            ratio = M.STAGE_MAPPING_RATIO[s]
            loss_grid = ops.grid_bce_loss(logits, rois, gt_boxes, ev.whole_map_size, ev.sub_regions, ratio,
                                          ev.pos_radius, ev.loss_weight)
            losses["loss_grid_%d" % (s + 1)] = loss_grid * self.stage_loss_weight[s]
            if G.IOU_HELPER and last:
                iou_target = torch.stack([1 - max_iou, max_iou], dim=1)
                losses["loss_iou_%d" % (s + 1)] = ops.l2_loss_nosync(iou_logits, iou_target) * G.IOU_LOSS_WEIGHT
            if last:
                break
            with torch.no_grad():
                # GridPostProcessor.forward(is_train=True) + the next stage's subsample: drop RoIs coinciding with
                # a gt, decode the rest, append the gts, keep what matches a gt at the next stage's IoU
                refined, keep = ops.grid_decode(logits, rois, ev.whole_map_size, ev.sub_regions, ratio, img, gt_all,
                                                gt_off)
                m2, iou2 = ops.match_rois(refined, img, gt_all, gt_off, M.FG_IOU_THRESHOLD[s + 1],
                                          M.BG_IOU_THRESHOLD[s + 1])
                keep_h = (keep & (m2 >= 0)).cpu().numpy()                       # the stage's one host round trip
                off = np.concatenate([[0], np.cumsum(counts)])
                n_roi, index, counts = int(off[-1]), [], []
                for i in range(n_img):
                    kept = np.flatnonzero(keep_h[off[i]:off[i + 1]]) + off[i]
                    index.extend(kept.tolist())
                    index.extend(range(n_roi + int(gt_off_h[i]), n_roi + int(gt_off_h[i + 1])))
                    counts.append(len(kept) + gt_counts[i])
                index = _to_device_async(index, torch.int64, dev)
                # every appended gt matches itself with IoU exactly 1 (inter == area)
                rois = torch.cat([refined, gt_all], dim=0)[index]
                gt_boxes = torch.cat([gt_all[m2.clamp(min=0) + base], gt_all], dim=0)[index]
                max_iou = torch.cat([iou2, torch.ones_like(gt_all[:, 0])], dim=0)[index]
                src = torch.cat([src, gt_src], dim=0)[index]
                img, base = roi_img_and_base(counts)
        result, o = [], 0
        lab, ob = labels[src], obj[src]
        for i in range(n_img):
            bl = BoxList(rois[o:o + counts[i]], sizes[i], mode="xyxy")
            bl.add_field("objectness", ob[o:o + counts[i]])
            bl.add_field("labels", lab[o:o + counts[i]])
            result.append(bl)
            o += counts[i]
        return x, result, losses

    def _forward_train_lists(self, features, props, targets):
        """The training forward of forward() above with every RoI set built on the device (pet/lib/ops/roi_lists.py):
        one launch samples the cls RoIs and lists their positives (CLSLossComputation.subsample +
        keep_only_positive_boxes), one launch per stage transition filters the decoded boxes, appends the gts and
        re-matches (GridPostProcessor + GridLossComputation.subsample), one gathers the RSM candidates
        (get_full_sample_boxes) and one samples them.  The cls and RSM heads run on the whole capacity of their
        sample (images x BATCH_SIZE_PER_IMAGE rows; the rows beyond the count carry cross_entropy's ignore_index
        and contribute exactly nothing), so only the grid stages need their RoI counts on the host: three small
        copies per step, the first hidden behind the cls head.  No index list is built on the host."""
        G, M = cfg.GRID_RCNN, cfg.GRID_RCNN.CASCADE_MAPPING_OPTION
        self._resolve_pending()
        # the five heads pool the same pyramid: their RoIAlign gradients are formed by one pass over it (pooler_fpn)
        ops.roi_backward_group(features)
        n_img, sizes = props.n_img, props.sizes
        gt_all, gt_labels, gt_off, off_h = RL.gt_pack(targets)
        n_gt = off_h[-1]
        ev = self.cls_loss_evaluator
        m, sp = ev.proposal_matcher, ev.fg_bg_sampler
        seeds = torch.randint(0, 2 ** 62, (3,)).tolist()              # CPU generator: torch.manual_seed reproduces
        with torch.no_grad():
            sample, pos, counts_all = RL.roi_sample(props, gt_all, gt_labels, gt_off, m.high_threshold,
                                                    m.low_threshold, sp.batch_size_per_image, sp.positive_fraction,
                                                    seeds[0], self.max_sample_num_grid, seeds[1],
                                                    M.FG_IOU_THRESHOLD[0])
            self._count_reads[0].start(counts_all)
        # ---- cls head on the sample's capacity, queued before the host waits for the counts -----------------
        # Its kernels (RoIAlign: HBM-bound; fc6 / fc7: 128-tile GEMMs) go to the second stream, beside the first grid
        # stage's convolutions on ~100 RoIs (ops.fwd_fork); the loss -- framework ops, which follow torch's current
        # stream -- is formed behind the join at the end.  The RSM head does the same beside the last grid stage.
        loss = {}
        cls_forked = ops.fwd_fork(features[0], 8)
        if cls_forked:
            with ops.fwd_side(features[0]):
                cls_logits = _head_logits(self.Head_cls, self.Output_cls, features, _capacity_rows(sample))
        else:
            ev.set_packed_sample(None, sample.labels)
            loss["loss_classifier"] = ev([_head_logits(self.Head_cls, self.Output_cls, features, _capacity_rows(sample))])
        c = self._count_reads[0].wait()                               # host round trip 1 of 3
        if c[-1]:
            raise RuntimeError("an image holds more than %d proposals; set CPM_DEVICE_LISTS=0" % RL.roi_sample_max_rows())
        sample.host_counts, pos.host_counts = c[:n_img + 1], c[n_img + 1:2 * (n_img + 1)]
        S = sample.total
        self._last_counts["cls"] = S
        ev.set_packed_sample(_RowsNow(sample, sizes, (("objectness", "obj"), ("labels", "labels"))), sample.labels)
        # ---- CMM cascade ---------------------------------------------------------------------------------
        cur, R0 = pos, pos.total
        x = None
        rsm = None
        for s in range(self.stage_num):
            gev = self.grid_loss_evaluators[s]
            last = s == self.stage_num - 1
            if last and G.RESCORE_ON:
                # the RSM sample needs the boxes this last stage STARTS from: its head runs beside the stage
                rsm = self._rsm_head(features, sample, pos, cur, R0, S, gt_all, gt_labels, gt_off, seeds[2], sizes)
            R = cur.total
            self._last_counts["grid_%d" % s] = R
            rois = cur.boxes[:R]
            x, _ = getattr(self, "Head_grid_%d" % s)(features, _RowsNow(cur, sizes))
            grid_logits, iou_logits = getattr(self, "Output_grid_%d" % s)(x, None)
            logits = grid_logits["unfused"]
            ratio = M.STAGE_MAPPING_RATIO[s]
            # the stage's weight rides on the loss kernel's own weight (value and gradient leave it scaled): no
            # elementwise kernel forward, none backward, no autograd node -- all of them queued where the device waits
            # for the host, the start of the backward pass.  (A weight of exactly 1 elsewhere is not multiplied either.)
            loss["loss_grid_%d" % (s + 1)] = ops.grid_bce_loss(logits, rois, cur.gt[:R], gev.whole_map_size,
                                                              gev.sub_regions, ratio, gev.pos_radius,
                                                              gev.loss_weight * float(self.stage_loss_weight[s]))
            if G.IOU_HELPER and last:
                loss["loss_iou_%d" % (s + 1)] = _scaled(ops.l2_loss_fused(iou_logits, iou=cur.iou[:R]), G.IOU_LOSS_WEIGHT)
            if last:
                break
            with torch.no_grad():
                img = cur.img[:R]
                refined, keep = ops.grid_decode(logits, rois, gev.whole_map_size, gev.sub_regions, ratio, img, gt_all,
                                                gt_off)
                m2, iou2 = ops.match_rois(refined, img, gt_all, gt_off, M.FG_IOU_THRESHOLD[s + 1],
                                          M.BG_IOU_THRESHOLD[s + 1])
                nxt = RL.stage_advance(refined, keep, m2, iou2, img, cur.src[:R] if s else None, n_img, R0, gt_all,
                                       gt_off, n_gt, sizes)
                self._count_reads[1 + s].start(nxt.counts)
                nxt.host_counts = self._count_reads[1 + s].wait()     # the stage's one host round trip
                cur = nxt
        if cls_forked or (rsm is not None and rsm[3]):
            ops.fwd_join(features[0])                                 # the compute stream has the heads' logits now
        if cls_forked:
            ev.set_packed_sample(None, sample.labels)
            loss["loss_classifier"] = ev([cls_logits])
            ev.set_packed_sample(_RowsNow(sample, sizes, (("objectness", "obj"), ("labels", "labels"))), sample.labels)
        if not G.RESCORE_ON:
            return x, _views(cur, sizes), loss
        result, rs, logits, _ = rsm
        rev = self.rescore_loss_evaluator
        rev.set_packed_sample(result, rs.labels)
        loss["loss_rescore"] = _scaled(rev([logits]), G.RESCORE_LOSS_WEIGHT)
        return x, result, loss

    def _rsm_head(self, features, sample, pos, cur, R0, S, gt_all, gt_labels, gt_off, seed, sizes):
        """RSM on the sample's capacity (its count is read back lazily): candidates = the cls sample's negatives + the
        boxes the last stage starts from, one sampling launch, then the head -- on the second stream when there is one.
        Returns (lazy views, the sampled list, logits, forked)."""
        rev = self.rescore_loss_evaluator
        m, sp = rev.proposal_matcher, rev.fg_bg_sampler
        with torch.no_grad():
            if cur is pos:                                            # single-stage cascade: rows are their own source
                cur.src = torch.arange(R0, device=cur.boxes.device)
            cand = RL.rescore_gather(sample, cur, pos.src, R0, S + cur.total)
            rs, _, counts_all = RL.roi_sample(cand, gt_all, gt_labels, gt_off, m.high_threshold, m.low_threshold,
                                              sp.batch_size_per_image, sp.positive_fraction, seed)
            self._count_reads[-1].start(counts_all)
        result = _LazyViews(self, rs, sizes)
        self._pending = result
        forked = ops.fwd_fork(features[0], 16)
        if forked:
            with ops.fwd_side(features[0]):
                logits = _head_logits(self.Head_rescore, self.Output_rescore, features, _capacity_rows(rs))
        else:
            logits = _head_logits(self.Head_rescore, self.Output_rescore, features, _capacity_rows(rs))
        return result, rs, logits, forked

    def _resolve_pending(self):
        """read the RSM sample's counts of the previous step (and its status) if nobody asked for them yet"""
        if self._pending is not None:
            self._pending.resolve()

    @property
    def last_counts(self):
        self._resolve_pending()
        return self._last_counts

    def _forward_train_rescore(self, features, cls_proposals, grid_proposals, targets):
        with torch.no_grad():
            proposals = get_full_sample_boxes(cls_proposals, grid_proposals)
            proposals = self.rescore_loss_evaluator.subsample(proposals, targets)
        self.last_counts["rescore"] = sum(len(p) for p in proposals)
        logits = self.Output_rescore(self.Head_rescore(features, proposals))
        return proposals, dict(loss_rescore=self.rescore_loss_evaluator([logits]) * cfg.GRID_RCNN.RESCORE_LOSS_WEIGHT)

    # ---- testing -----------------------------------------------------------------------------------------
    def _forward_test_cls(self, features, proposals):
        logits = self.Output_cls(self.Head_cls(features, proposals))
        return self.cls_post_processor(logits, proposals)

    def _forward_test_cascade(self, features, proposals):
        x = None
        for s in range(self.stage_num):
            x, _ = getattr(self, "Head_grid_%d" % s)(features, proposals)
            grid_logits, iou_logits = getattr(self, "Output_grid_%d" % s)(x, None)
            if s == self.stage_num - 1 and self.test_ensemble:
                raise Exception("unsupported operation!")
            proposals = self.grid_post_processors[s](grid_logits, proposals, iou_logits)
            if s < self.stage_num - 1 and s == self.test_stage - 1:
                break
        return x, proposals

    def _forward_test_rescore(self, features, proposals):
        logits = self.Output_rescore(self.Head_rescore(features, proposals))
        return self.cls_post_processor(logits, proposals, rescore=True)


def _scaled(loss, weight):
    return loss if float(weight) == 1.0 else loss * weight


def _head_logits(head, output, features, rows):
    """RoIAlign -> fc6 -> ReLU -> fc7 -> ReLU -> cls_score of a cls / RSM head: the three Linears as one autograd node
    (pet/lib/ops/conv.py: mlp_chain); the backward pass starts with these, and op by op its first kernels are the
    smallest of the step"""
    return conv_ops.mlp_chain(head.pooler(features, rows), [head.fc6, head.fc7, output.cls_score], head)


class _PackedBoxLists(list):
    """per-image BoxLists that are views of one packed device list, plus its [R, 5] RoIAlign rows"""
    rois5 = None


class _RowsNow(object):
    """the live rows of a packed list for a head: the [R, 5] RoIAlign rows at once, the per-image BoxList views only
    if somebody iterates (hooks, tests) -- building them costs host time right behind a host round trip"""

    def __init__(self, lst, sizes, fields=()):
        self._lst, self._sizes, self._fields, self._lists = lst, sizes, fields, None
        self.rois5 = lst.rois5[:lst.total] if lst.rois5 is not None else None

    def _get(self):
        if self._lists is None:
            self._lists = _views(self._lst, self._sizes, *self._fields)
        return self._lists

    def __len__(self):
        return self._lst.n_img

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, i):
        return self._get()[i]


def _capacity_rows(lst):
    """all `capacity` rows of a sampled list as the RoIAlign input of a head that runs without knowing the count"""
    out = _PackedBoxLists()
    out.rois5 = lst.rois5
    return out


class _LazyViews(object):
    """the RSM sample as per-image BoxLists, materialised (one small device->host copy, queued long before) only when
    somebody looks at it: the training loop never does"""

    def __init__(self, head, lst, sizes):
        self._head, self._lst, self._sizes, self._lists = head, lst, sizes, None

    def resolve(self):
        if self._lists is None:
            head, n = self._head, self._lst.n_img
            c = head._count_reads[-1].wait()
            if head._pending is self:
                head._pending = None
            if c[-1]:
                raise RuntimeError("an image holds more than %d RSM candidates; set CPM_DEVICE_LISTS=0"
                                   % RL.roi_sample_max_rows())
            self._lst.host_counts = c[:n + 1]
            head._last_counts["rescore"] = self._lst.total
            self._lists = _views(self._lst, self._sizes, ("objectness", "obj"), ("labels", "labels"))
        return self._lists

    def __len__(self):
        return len(self.resolve())

    def __iter__(self):
        return iter(self.resolve())

    def __getitem__(self, i):
        return self.resolve()[i]


def _views(lst, sizes, *fields):
    """RoIList -> per-image BoxList views (no device work); fields: (BoxList field, RoIList attribute) pairs"""
    out, o = _PackedBoxLists(), 0
    for i in range(lst.n_img):
        c = lst.host_counts[i]
        bl = BoxList(lst.boxes[o:o + c], sizes[i], mode="xyxy")
        for name, attr in fields:
            bl.add_field(name, getattr(lst, attr)[o:o + c])
        out.append(bl)
        o += c
    out.rois5 = lst.rois5[:o] if lst.rois5 is not None else None
    return out


def get_full_sample_boxes(cls_proposals, grid_proposals):
    """RSM sample = negatives of the cls sample + the refined positives (grid_cascade_rcnn.py:231-245)."""
    out = []
    for c, g in zip(cls_proposals, grid_proposals):
        hl = getattr(c, "host_labels", None)
        if hl is not None and len(hl) == len(c):
            inds = torch.from_numpy(np.flatnonzero(hl <= 0)).pin_memory().to(c.bbox.device, non_blocking=True)
        else:
            inds = (c.get_field("labels") <= 0).nonzero().squeeze(1)
        if cfg.GRID_RCNN.RESCORE_OPTION.KEEP_RATIO:
            neg_num = g.bbox.shape[0] * 3
            if neg_num <= inds.shape[0]:
                inds = inds[torch.randperm(inds.shape[0], device=inds.device)[:neg_num]]
        out.append(cat_boxlist((c[inds], g)))
    return out
