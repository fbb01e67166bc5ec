"""Device-resident RoI lists of the training step: host wrappers of cpm_proposals_finalize / cpm_roi_sample /
cpm_stage_advance / cpm_rescore_gather (include/cpmrcnn_hip.h, csrc/roi_lists.hip).

A list is a set of capacity-sized tensors whose first `total` rows are live, image-contiguous, with the per-image row
counts held on the DEVICE ([images] counts followed by the total).  The reference's BoxList surgery between the
proposal NMS and the heads (pet/rcnn/modeling/rpn/inference.py:101-196, grid_cascade_rcnn/loss.py:29-97,
pet/rcnn/utils/misc.py:54-94, grid_cascade_rcnn.py:231-245) becomes one launch per list; the host only reads the
counts back (`Counts.start` / `Counts.wait`) when it must size the next head launch."""
import ctypes
import os

import torch

from . import _hip as H


class RoIList(object):
    """Packed rows [capacity, ...]; `counts` int32 [images + 1] on the device (per image, then total)."""
    __slots__ = ("boxes", "obj", "img", "rois5", "labels", "src", "gt", "iou", "counts", "capacity", "n_img", "sizes",
                 "host_counts")

    def __init__(self, capacity, n_img, sizes=None, **fields):
        self.capacity, self.n_img, self.sizes = capacity, n_img, sizes
        self.host_counts = None
        for k in ("boxes", "obj", "img", "rois5", "labels", "src", "gt", "iou", "counts"):
            setattr(self, k, fields.get(k))

    @property
    def total(self):
        if self.host_counts is None:
            raise RuntimeError("RoIList counts have not been read back (Counts.start / Counts.wait)")
        return self.host_counts[-1]

    def to_boxlists(self):
        """per-image BoxLists (bbox + "objectness"), reading the counts back if needed -- for consumers outside the
        packed training path (tests, tools)"""
        from pet.utils.data.structures.bounding_box import BoxList
        if self.host_counts is None:
            self.host_counts = self.counts.cpu().tolist()
        out, o = [], 0
        for i in range(self.n_img):
            c = self.host_counts[i]
            bl = BoxList(self.boxes[o:o + c], self.sizes[i], mode="xyxy")
            if self.obj is not None:
                bl.add_field("objectness", self.obj[o:o + c])
            out.append(bl)
            o += c
        return out

    def live(self, name):
        """the live rows of a field (a view); needs the counts on the host"""
        return getattr(self, name)[:self.total]


_COUNTS_MAPPED = os.environ.get("CPM_COUNTS_MAPPED", "1") != "0"


class Counts(object):
    """Device counts -> host, one per call site.  Default: ONE small launch stores the counts and then a sequence number
    into pinned host memory the device has mapped (cpm_publish_counts) and the host polls the sequence word -- no copy
    command, no event wait, no wake-up.  An event is recorded behind the launch all the same: a poll that has not seen
    the word after two seconds waits for the event instead (visibility is then the runtime's business).
    CPM_COUNTS_MAPPED=0: a copy into a pinned staging buffer and an event."""

    def __init__(self):
        self._pin = None
        self._event = None
        self._np = None
        self._dptr = None
        self._seq = 0

    def _alloc(self, n):
        self._pin = torch.zeros(max(n + 1, 64), dtype=torch.int32).pin_memory()
        self._np = self._pin.numpy()
        self._dptr = None
        if _COUNTS_MAPPED:
            out = ctypes.c_void_p()
            rc = H.lib().cpm_host_device_pointer(ctypes.c_void_p(self._pin.data_ptr()), ctypes.byref(out))
            self._dptr = out if rc == 0 and out.value else None      # (not mapped on this system: the copy path)

    def start(self, dev_counts):
        n = dev_counts.numel()
        if self._pin is None or self._pin.numel() < n + 1:
            self._alloc(n)
        self._n = n
        if self._event is None:
            self._event = torch.cuda.Event()
        if self._dptr is not None and dev_counts.dtype == torch.int32 and dev_counts.is_contiguous() and n <= 255:
            self._seq = (self._seq % 0x3fffffff) + 1
            self._np[n] = 0
            with H.guard(dev_counts.device):
                rc = H.lib().cpm_publish_counts(H.ptr(dev_counts), n, self._dptr, self._seq, H.stream())
            H.check(rc, "publish_counts")
            self._polled = True
        else:
            self._pin[:n].copy_(dev_counts, non_blocking=True)
            self._polled = False
        self._event.record()

    def wait(self):
        if self._polled:
            a, n, seq = self._np, self._n, self._seq
            spins = 0
            while a[n] != seq:
                spins += 1
                if not (spins & 0xfff) and self._event.query():
                    # the launch is complete: whatever the stores' path, they are visible behind the event
                    self._event.synchronize()
                    break
            return a[:n].tolist()
        self._event.synchronize()
        return self._np[:self._n].tolist()


def gt_pack(targets):
    """(gt boxes [G,4], gt labels int64 [G], gt_off int32 [images+1] device, host offsets) of a batch, built once per
    step and remembered on the first target."""
    first = targets[0]
    key = tuple((id(t), t.bbox.data_ptr(), t.bbox._version, len(t)) for t in targets)
    pack = getattr(first, "_cpm_gt_pack", None)
    if pack is not None and pack[4] == key:
        return pack[:4]
    dev = first.bbox.device
    off_h = [0]
    for t in targets:
        off_h.append(off_h[-1] + len(t))
    gt_all = torch.cat([t.bbox for t in targets], dim=0).contiguous()
    labels = torch.cat([t.get_field("labels") for t in targets], dim=0).to(torch.int64) \
        if first.has_field("labels") else None
    gt_off = torch.tensor(off_h, dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
    first._cpm_gt_pack = (gt_all, labels, gt_off, off_h, key)
    return gt_all, labels, gt_off, off_h


def proposals_finalize(seg_boxes, seg_scores, keep, keep_count, seg_off, n_img, n_lvl, post_nms_top_n, batch_top_k,
                       gt_all, gt_off, n_gt, sizes):
    H.require_gpu(seg_boxes, seg_scores, gt_all)
    dev = seg_boxes.device
    cand = sum(min(seg_off[s + 1] - seg_off[s], post_nms_top_n) if post_nms_top_n > 0 else seg_off[s + 1] - seg_off[s]
               for s in range(n_img * n_lvl))
    cap = min(int(batch_top_k), cand) + int(n_gt)
    out = RoIList(cap, n_img, sizes,
                  boxes=torch.empty((cap, 4), dtype=torch.float32, device=dev),
                  obj=torch.empty((cap,), dtype=torch.float32, device=dev),
                  img=torch.empty((cap,), dtype=torch.int32, device=dev),
                  counts=torch.empty((n_img + 1,), dtype=torch.int32, device=dev))
    offs = (ctypes.c_int32 * (n_img * n_lvl + 1))(*seg_off)
    with H.guard(dev):
        rc = H.lib().cpm_proposals_finalize(H.ptr(seg_boxes), H.ptr(seg_scores), H.ptr(keep), H.ptr(keep_count), offs,
                                            n_img, n_lvl, int(post_nms_top_n), int(batch_top_k), H.ptr(gt_all),
                                            H.ptr(gt_off), cap, H.ptr(out.boxes), H.ptr(out.obj), H.ptr(None),
                                            H.ptr(out.img), H.ptr(out.counts), H.stream())
    H.check(rc, "proposals_finalize")
    return out


_MAX_ROWS = None


def roi_sample_max_rows():
    global _MAX_ROWS
    if _MAX_ROWS is None:
        _MAX_ROWS = int(H.lib().cpm_roi_sample_max_rows())
    return _MAX_ROWS


def roi_sample(inp, gt_all, gt_labels, gt_off, high, low, batch, positive_fraction, seed, max_grid=0, seed_grid=0,
               grid_high=0.5):
    """inp: RoIList(boxes, obj, counts).  Returns (sample RoIList, positives RoIList or None, counts_all): counts_all
    int32 [2 * (images + 1) + 1] = sample counts, positives counts, status -- one tensor so that one copy reads it."""
    dev = inp.boxes.device
    n = inp.n_img
    # An image's rows are bounded by the list's capacity: checked HERE, before anything is queued -- the kernel's own
    # overflow status is only read back after the cls head has been launched on the sample it would have left unwritten.
    if inp.capacity > roi_sample_max_rows():
        raise RuntimeError("a list of capacity %d may hold more rows per image than cpm_roi_sample takes (%d): "
                           "set CPM_DEVICE_LISTS=0" % (inp.capacity, roi_sample_max_rows()))
    max_pos = int(batch * positive_fraction)
    cap_s = n * int(batch)
    cap_g = n * min(int(max_grid), max_pos) if max_grid > 0 else 0
    counts_all = torch.empty((2 * (n + 1) + 1,), dtype=torch.int32, device=dev)
    s = RoIList(cap_s, n, inp.sizes,
                boxes=torch.empty((cap_s, 4), dtype=torch.float32, device=dev),
                obj=torch.empty((cap_s,), dtype=torch.float32, device=dev),
                labels=torch.empty((cap_s,), dtype=torch.int64, device=dev),
                img=torch.empty((cap_s,), dtype=torch.int32, device=dev),
                rois5=torch.empty((cap_s, 5), dtype=torch.float32, device=dev),
                counts=counts_all[:n + 1])
    p = None
    if max_grid > 0:
        p = RoIList(cap_g, n, inp.sizes,
                    boxes=torch.empty((cap_g, 4), dtype=torch.float32, device=dev),
                    gt=torch.empty((cap_g, 4), dtype=torch.float32, device=dev),
                    iou=torch.empty((cap_g,), dtype=torch.float32, device=dev),
                    src=torch.empty((cap_g,), dtype=torch.int64, device=dev),
                    img=torch.empty((cap_g,), dtype=torch.int32, device=dev),
                    rois5=torch.empty((cap_g, 5), dtype=torch.float32, device=dev),
                    counts=counts_all[n + 1:2 * (n + 1)])
    status = counts_all[2 * (n + 1):]
    with H.guard(dev):
        rc = H.lib().cpm_roi_sample(
            H.ptr(inp.boxes), H.ptr(inp.obj), H.ptr(inp.counts), n, H.ptr(gt_all), H.ptr(gt_labels), H.ptr(gt_off),
            H.f(high), H.f(low), int(batch), max_pos, ctypes.c_uint64(int(seed)), int(max_grid),
            ctypes.c_uint64(int(seed_grid)), H.f(grid_high), cap_s, H.ptr(s.boxes), H.ptr(s.obj), H.ptr(s.labels),
            H.ptr(s.img), H.ptr(s.rois5), H.ptr(s.counts), cap_g, H.ptr(p.boxes if p else None),
            H.ptr(p.gt if p else None), H.ptr(p.iou if p else None), H.ptr(p.src if p else None),
            H.ptr(p.img if p else None), H.ptr(p.rois5 if p else None), H.ptr(counts_all[n + 1:]),
            H.ptr(status), H.stream())
    H.check(rc, "roi_sample")
    return s, p, counts_all


def stage_advance(refined, keep, matched, iou, img, src, n_img, gt_src_base, gt_all, gt_off, n_gt, sizes):
    dev = refined.device
    R = refined.shape[0]
    cap = R + int(n_gt)
    keep = keep if keep.dtype == torch.uint8 else keep.view(torch.uint8)
    out = RoIList(cap, n_img, sizes,
                  boxes=torch.empty((cap, 4), dtype=torch.float32, device=dev),
                  gt=torch.empty((cap, 4), dtype=torch.float32, device=dev),
                  iou=torch.empty((cap,), dtype=torch.float32, device=dev),
                  src=torch.empty((cap,), dtype=torch.int64, device=dev),
                  img=torch.empty((cap,), dtype=torch.int32, device=dev),
                  rois5=torch.empty((cap, 5), dtype=torch.float32, device=dev),
                  counts=torch.empty((n_img + 1,), dtype=torch.int32, device=dev))
    with H.guard(dev):
        rc = H.lib().cpm_stage_advance(H.ptr(refined), H.ptr(keep), H.ptr(matched), H.ptr(iou), H.ptr(img), H.ptr(src),
                                       R, n_img, ctypes.c_int64(int(gt_src_base)), H.ptr(gt_all), H.ptr(gt_off), cap,
                                       H.ptr(out.boxes), H.ptr(out.gt), H.ptr(out.iou), H.ptr(out.src), H.ptr(out.img),
                                       H.ptr(out.rois5), H.ptr(out.counts), H.stream())
    H.check(rc, "stage_advance")
    return out


def rescore_gather(sample, last, first_src, n_first, capacity):
    """sample: the cls sample (boxes, obj, labels, counts); last: the last stage's list (boxes, src, counts);
    first_src: row in `sample` of each first-stage RoI (n_first of them).  -> RoIList(boxes, obj, counts)"""
    dev = last.boxes.device
    n = sample.n_img
    out = RoIList(capacity, n, sample.sizes,
                  boxes=torch.empty((capacity, 4), dtype=torch.float32, device=dev),
                  obj=torch.empty((capacity,), dtype=torch.float32, device=dev),
                  counts=torch.empty((n + 1,), dtype=torch.int32, device=dev))
    with H.guard(dev):
        rc = H.lib().cpm_rescore_gather(H.ptr(sample.boxes), H.ptr(sample.obj), H.ptr(sample.labels),
                                        H.ptr(sample.counts), H.ptr(last.boxes), H.ptr(last.src), H.ptr(last.counts),
                                        H.ptr(first_src), int(n_first), n, int(capacity), H.ptr(out.boxes),
                                        H.ptr(out.obj), H.ptr(out.counts), H.stream())
    H.check(rc, "rescore_gather")
    return out


def from_boxlists(boxlists):
    """per-image BoxLists (bbox + "objectness") -> the packed list the RPN's device path would hand over"""
    dev = boxlists[0].bbox.device
    counts = [len(b) for b in boxlists]
    img = torch.cat([torch.full((c,), i, dtype=torch.int32, device=dev) for i, c in enumerate(counts)])
    out = RoIList(sum(counts), len(boxlists), [b.size for b in boxlists],
                  boxes=torch.cat([b.bbox for b in boxlists], dim=0).contiguous(),
                  obj=torch.cat([b.get_field("objectness") for b in boxlists], dim=0).contiguous(), img=img,
                  counts=torch.tensor(counts + [sum(counts)], dtype=torch.int32, device=dev))
    out.host_counts = counts + [sum(counts)]
    return out
