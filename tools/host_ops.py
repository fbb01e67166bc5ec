#!/usr/bin/env python3
"""Host-side cost of the RoI-head section of a training step (the stretch between the proposal count's round trip
and the start of backward, where the device waits for the host): top-level host ops by total time and count."""
import collections
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
for _ in range(4):
    tr.step(images, targets)
torch.cuda.synchronize()

heads = tr.model._roi_heads()
orig = heads.forward


def wrapped(*a, **k):
    with record_function("ROI_HEADS_FORWARD"):
        return orig(*a, **k)


heads.forward = wrapped
N = 3
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(N):
        tr.step(images, targets)
    torch.cuda.synchronize()
evs = [e for e in prof.events()]
spans = [(e.time_range.start, e.time_range.end) for e in evs if e.name == "ROI_HEADS_FORWARD"]
print("RoI heads forward: %.2f ms host per step" % (sum(b - a for a, b in spans) / N / 1e3))
agg = collections.defaultdict(lambda: [0, 0.0])
covered = 0.0
for e in evs:
    if e.name == "ROI_HEADS_FORWARD" or e.cpu_parent is None or e.cpu_parent.name != "ROI_HEADS_FORWARD":
        continue
    agg[e.name][0] += 1
    agg[e.name][1] += e.cpu_time_total
    covered += e.cpu_time_total
print("ops cover %.2f ms per step; the rest is Python between them" % (covered / N / 1e3))
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us/step  x%-6.1f %s" % (t / N, n / N, name[:100]))
