#!/usr/bin/env python3
"""Which host-side call sites launch the large framework copy / fill / add kernels of a training step (with stacks)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
tr = bench.Trainer(dev)
images, targets = bench.synthetic_batch(2, 800, 1333, 16, 1234, dev)
cal, _ = bench.synthetic_batch(2, 800, 1333, 1, 4321, dev)
bench.calibrate_frozen_affine(tr.model, cal.tensors)
for _ in range(4):
    tr.step(images, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        tr.step(images, targets)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type.name != "CPU" or not e.name.startswith("aten::"):
        continue
    dt = getattr(e, "device_time_total", 0) or 0
    self_dt = getattr(e, "self_device_time_total", 0) or 0
    if self_dt < float(os.environ.get("MIN_US", "8")):       # us of kernel time attributed to this op itself
        continue
    st = [s for s in (e.stack or []) if "cpm-r-cnn_amd" in s or "bench.py" in s]
    key = (e.name, str(e.input_shapes)[:80], st[0][-90:] if st else "?")
    agg[key][0] += 1
    agg[key][1] += self_dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("ROWS", "45"))]
for (name, shp, where), (n, t) in rows:
    print("%7.1f us/step x%-4.1f %-22s %-60s %s" % (t / 2, n / 2, name, shp, where))
