#!/usr/bin/env python3
"""Host time per phase of the training step (perf_counter stamps, no sync added) next to the device time of the same
phases (CUDA events on the compute stream): where does the device wait for the host?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
from pet.lib.ops import _hip  # noqa: E402
_hip.set_conv_math("bf16x3")
tr = bench.Trainer(dev)
images, targets = bench.synthetic_batch(2, 800, 1333, 16, 1234, dev)
cal, _ = bench.synthetic_batch(2, 800, 1333, 1, 4321, dev)
bench.calibrate_frozen_affine(tr.model, cal.tensors)
from pet.utils.parallel import backward_losses  # noqa: E402
from pet.utils.data.structures.image_list import to_image_list  # noqa: E402
m = tr.model
names = ["features", "rpn", "heads", "backward", "optimizer"]


def step(stamps, events):
    def mark(i):
        stamps[i].append(time.perf_counter())
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        events[i].append(e)
    tr.scheduler.step()
    tr.optimizer.zero_grad()
    tr.reducer.begin_step()
    mark(0)
    il = to_image_list(images)
    feats = m._features(il.tensors)
    mark(1)
    proposals, pl = m.RPN(il, feats, targets)
    mark(2)
    _, _, rl = m._roi_heads()(feats, proposals, targets)
    mark(3)
    losses = dict(pl)
    losses.update(rl)
    backward_losses(losses)
    mark(4)
    tr.reducer.finish()
    tr.optimizer.step()
    mark(5)


for _ in range(5):
    step([[] for _ in range(6)], [[] for _ in range(6)])
torch.cuda.synchronize()
n = 20
stamps, events = [[] for _ in range(6)], [[] for _ in range(6)]
t0 = time.perf_counter()
for _ in range(n):
    step(stamps, events)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("step %.2f ms" % ((t1 - t0) * 1e3 / n))
print("%-10s %12s %12s" % ("phase", "host ms", "device ms"))
for i, nm in enumerate(names):
    h = sum(stamps[i + 1][k] - stamps[i][k] for k in range(n)) / n * 1e3
    d = sum(events[i][k].elapsed_time(events[i + 1][k]) for k in range(n)) / n
    print("%-10s %12.2f %12.2f" % (nm, h, d))
# how far ahead of the device is the host at each mark?  (device timestamp of mark i minus host timestamp, relative)
