#!/usr/bin/env python3
"""Host time (perf_counter, no device sync added) of the sections of the RoI-head window of a training step -- the
stretch in which the device waits for the host (tools/step_gaps.py) -- to see which part of the Python path holds it up."""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
tr = bench.Trainer(dev)
images, targets = bench.synthetic_batch(2, 800, 1333, 16, 1234, dev)
cal, _ = bench.synthetic_batch(2, 800, 1333, 1, 4321, dev)
bench.calibrate_frozen_affine(tr.model, cal.tensors)
acc = collections.OrderedDict()


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
    setattr(obj, name, g)


m = tr.model
G = m.Grid_Cascade_RCNN
wrap(m.RPN.box_selector_train, "start_fused", "rpn.start_fused (topk, decode, nms launch)")
wrap(m.RPN.box_selector_train, "finish_fused", "rpn.finish_fused (sync 1+2, index lists)")
wrap(G.cls_loss_evaluator, "subsample", "cls.subsample (match, sampler, sync 3)")
wrap(G.rescore_loss_evaluator, "subsample", "rescore.subsample (sync 6)")
wrap(G, "_forward_train_cls", "cls total")
wrap(G, "_forward_train_cascade", "cascade total")
wrap(G, "_forward_train_rescore", "rescore total")
for s in range(3):
    wrap(getattr(G, "Head_grid_%d" % s), "forward", "  grid head %d (RoIAlign + 8 conv/GN)" % s)
    wrap(getattr(G, "Output_grid_%d" % s), "forward", "  grid output %d (2 deconv + GN [+ ISM])" % s)
wrap(G.Head_cls, "forward", "  cls head (RoIAlign + fc6 + fc7)")
wrap(G.Head_rescore, "forward", "  rescore head")
import pet.lib.ops as ops  # noqa: E402
import pet.rcnn.modeling.grid_cascade_rcnn.grid_cascade_rcnn as gcr  # noqa: E402
for fn in ("grid_bce_loss", "grid_decode", "match_rois"):
    wrap(gcr.ops, fn, "  ops.%s" % fn)
wrap(gcr, "get_full_sample_boxes", "  get_full_sample_boxes")
wrap(m, "_roi_heads", "roi_heads lookup") if hasattr(m, "_roi_heads") else None
for _ in range(5):
    tr.step(images, targets)
torch.cuda.synchronize()
acc.clear()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.step(images, targets)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("step %.2f ms" % (dt * 1e3))
for k, v in acc.items():
    print("%8.3f ms/step  %s" % (v / n * 1e3, k))
