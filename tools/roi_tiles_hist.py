#!/usr/bin/env python3
"""How the RoIAlign backward (gather formulation: one workgroup per 8x8 tile of the gradient pyramid, csrc/roi_align.hip)
is loaded in the benchmark's step: per call the number of RoIs, of non-empty tiles, and the distribution of RoIs per tile.
Reads the binning pass's counters out of the call's workspace after each backward call of one training step.
    python tools/roi_tiles_hist.py"""
import os
import sys

os.environ["CPM_ROI_BWD_GROUP"] = "0"      # call by call: the counters of one call at a time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from bench import Trainer, synthetic_batch
    from pet.lib.ops import _hip as H
    from pet.lib.ops import pooler_fpn as PF
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    H.set_conv_math("bf16x3")
    tr = Trainer(dev)
    images, targets = synthetic_batch(2, 800, 1333, 16, 0, dev)
    for _ in range(3):
        tr.step(images, targets)
    torch.cuda.synchronize()
    orig = PF._RoIAlignFPN.backward
    calls = []

    def wrapped(ctx, *grads):
        grad_out = grads[0]
        out = orig(ctx, *grads)
        torch.cuda.synchronize()
        shapes = ctx.meta[-1]
        tiles = sum(int(s[0]) * ((int(s[2]) + 7) // 8) * ((int(s[3]) + 7) // 8) for s in shapes)
        ws = H.workspace(1, grad_out.device)
        cnt = ws[64:64 + 4 * tiles].cpu().numpy().view(np.int32).copy()
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        orig(ctx, *grads)                                        # timed repeat (accumulates twice: diagnostic only)
        t1.record()
        torch.cuda.synchronize()
        calls.append((int(ctx.saved_tensors[0].shape[0]), tuple(grad_out.shape[1:]), cnt, t0.elapsed_time(t1) * 1e3,
                      [tuple(int(v) for v in s) for s in shapes]))
        return out

    PF._RoIAlignFPN.backward = staticmethod(wrapped)
    tr.step(images, targets)
    torch.cuda.synchronize()
    for K, shp, cnt, us, shapes in calls:
        act = cnt[cnt > 0]
        print("K=%4d out %s: %6.1f us incl. fills | tiles %d, non-empty %d, (tile, RoI) pairs %d, per tile: mean %.1f "
              "p50 %d p90 %d p99 %d max %d" % (K, shp, us, cnt.size, act.size, int(act.sum()), act.mean(),
                                             np.percentile(act, 50), np.percentile(act, 90), np.percentile(act, 99),
                                             act.max()))
        base = 0
        for s in shapes:
            n = s[0] * ((s[2] + 7) // 8) * ((s[3] + 7) // 8)
            c = cnt[base:base + n]
            a = c[c > 0]
            print("      level %dx%d: non-empty %d of %d, pairs %d, max %d" % (s[2], s[3], a.size, n, int(a.sum()),
                                                                           int(a.max()) if a.size else 0))
            base += n


if __name__ == "__main__":
    main()
