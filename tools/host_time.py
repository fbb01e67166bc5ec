#!/usr/bin/env python3
"""How long the HOST needs to enqueue one training step (no device wait inside the loop except the step's own syncs)
next to the device-side step time: tells whether the step is launch bound."""
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
for _ in range(5):
    tr.step(images, targets)
torch.cuda.synchronize()
import cProfile, pstats  # noqa: E402
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.step(images, targets)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step; with final drain %.2f ms/step; cpus %d; load %s" % ((t1 - t0) * 1e3 / n, (t2 - t0) * 1e3 / n,
      os.cpu_count(), os.getloadavg()))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(images, targets)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
