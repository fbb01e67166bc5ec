#!/usr/bin/env python3
"""Times of the one-workgroup / few-workgroup kernels of the proposal and sampling chain at the benchmark's sizes
(2 x 268 569 anchors, 16 gts per image, 2 000 pre-NMS candidates per level and image), each call alone.
    python tools/time_glue.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    import pet.lib.ops as ops
    g = torch.Generator().manual_seed(3)
    dev = "cuda"
    sizes = [201600, 50400, 12600, 3150, 819]
    ss = [torch.sigmoid(torch.randn(2, n, generator=g) * 0.05).to(dev) for n in sizes]
    ks = [min(2000, n) for n in sizes]
    print("topk_rows_multi (5 levels x 2 rows): %7.1f us" % timeit(lambda: ops.topk_rows_multi(ss, ks)))
    print("topk_rows P2 alone (one workgroup per row): %7.1f us" % timeit(lambda: ops.topk_rows(ss[0], 2000)))
    # the anchors of an 800 x 1344 image as the generator lays them out: level by level, row by row, three shapes per cell
    per = []
    for stride, size in ((4, 32), (8, 64), (16, 128), (32, 256), (64, 512)):
        h, w = -(-800 // stride), -(-1344 // stride)
        ys, xs = torch.meshgrid(torch.arange(h) * stride, torch.arange(w) * stride, indexing="ij")
        ctr = torch.stack([xs, ys, xs, ys], -1).reshape(-1, 1, 4).float()
        shapes = torch.tensor([[-(size * r ** 0.5) / 2, -(size / r ** 0.5) / 2, (size * r ** 0.5) / 2, (size / r ** 0.5) / 2]
                               for r in (0.5, 1.0, 2.0)])
        per.append((ctr + shapes.view(1, 3, 4)).reshape(-1, 4))
    one = torch.cat(per, 0)
    R = 2 * one.shape[0]
    anchors = torch.cat([one, one], 0).contiguous().to(dev)
    img = torch.cat([torch.zeros(R // 2, dtype=torch.int32), torch.ones(R - R // 2, dtype=torch.int32)]).to(dev)
    gxy = torch.rand(32, 2, generator=g) * torch.tensor([1000., 600.])
    gts = torch.cat([gxy, gxy + torch.rand(32, 2, generator=g) * 300 + 20], 1).to(dev)
    gt_off = torch.tensor([0, 16, 32], dtype=torch.int32, device=dev)
    for low_q in (True, False):
        t = timeit(lambda: ops.match_rois(anchors, img, gts, gt_off, 0.7, 0.3, low_q))
        print("match_rois over %d anchors, allow_low_quality=%s: %7.1f us" % (R, low_q, t))


if __name__ == "__main__":
    main()
