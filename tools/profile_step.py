#!/usr/bin/env python3
"""torch.profiler view of one training step: which host-side ops launch the glue kernels (counts, CPU time)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile, record_function  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
tr = bench.Trainer(dev)
images, targets = bench.synthetic_batch(2, 800, 1333, 16, 1234, dev)
cal, _ = bench.synthetic_batch(2, 800, 1333, 1, 4321, dev)
bench.calibrate_frozen_affine(tr.model, cal.tensors)
for _ in range(5):
    tr.step(images, targets)
torch.cuda.synchronize()

m = tr.model
orig = {}


def _rf(label, f, *a, **k):
    with record_function(label):
        return f(*a, **k)


def wrap(obj, name, label):
    f = getattr(obj, name)
    orig[(obj, name)] = f

    def g(*a, **k):
        with record_function(label):
            return f(*a, **k)
    setattr(obj, name, g)


wrap(m, "_features", "SEC_backbone_fpn")
wrap(m.RPN, "forward", "SEC_rpn_total")
wrap(m.RPN.box_selector_train, "forward", "SEC_rpn_proposals")
m.RPN.loss_evaluator.__class__.__call__ = (lambda f: (lambda self, *a, **k: _rf("SEC_rpn_loss", f, self, *a, **k)))(m.RPN.loss_evaluator.__class__.__call__)
G = m.Grid_Cascade_RCNN
wrap(G, "_forward_train_cls", "SEC_cls")
wrap(G, "_forward_train_cascade", "SEC_cascade")
wrap(G, "_forward_train_rescore", "SEC_rescore")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        with record_function("SEC_step"):
            tr.step(images, targets)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
ev = [e for e in prof.key_averages() if e.key.startswith("SEC_")]
for e in ev:
    print("%-22s calls=%d cpu_total=%.2f ms/step  cuda_total=%.2f ms/step" % (e.key, e.count, e.cpu_time_total / 3e3,
                                                                            getattr(e, "device_time_total", 0) / 3e3))

# launches and synchronising copies per section (host-side view: which glue is launch bound)
evs = prof.events()
secs = [e for e in evs if e.name.startswith("SEC_") and e.device_type.name == "CPU"]
launch = [e for e in evs if e.name in ("hipLaunchKernel", "hipMemcpyWithStream", "hipMemsetAsync", "hipExtModuleLaunchKernel")]
import collections  # noqa: E402
per = collections.defaultdict(lambda: collections.Counter())
for s in secs:
    lo, hi = s.time_range.start, s.time_range.end
    for e in launch:
        if lo <= e.time_range.start <= hi:
            per[s.name][e.name] += 1
for k, v in per.items():
    print("%-22s %s" % (k, {a: round(b / 3, 1) for a, b in v.items()}))
ops_in = collections.defaultdict(lambda: collections.Counter())
aten = [e for e in evs if e.name.startswith("aten::") and e.device_type.name == "CPU"]
for s in secs:
    if s.name not in ("SEC_cascade", "SEC_rpn_proposals", "SEC_rpn_loss", "SEC_cls"):
        continue
    lo, hi = s.time_range.start, s.time_range.end
    for e in aten:
        if lo <= e.time_range.start <= hi:
            ops_in[s.name][e.name] += 1
for k, v in ops_in.items():
    print(k, [(a, round(b / 3)) for a, b in v.most_common(18)])
