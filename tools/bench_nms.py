"""Batched NMS at the RPN shape (10 segments x 2000 boxes, thr 0.7): wall time of the whole call (HIP events) and a
checksum of the kept lists.  Also usable under rocprofv3 --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpm-r-cnn_amd"))
import pet.lib.ops as ops  # noqa: E402


def run():
    gen = torch.Generator().manual_seed(0)
    n_seg, per = 10, 2000
    xy = torch.rand(n_seg * per, 2, generator=gen) * torch.tensor([1269., 736.])
    boxes = torch.cat([xy, xy + torch.rand(n_seg * per, 2, generator=gen) * 300 + 16], 1).cuda()
    scores = torch.rand(n_seg * per, generator=gen).cuda()
    offs = [i * per for i in range(n_seg + 1)]
    for _ in range(5):
        keep, counts = ops.nms_segments(boxes, scores, None, offs, 0.7, 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        keep, counts = ops.nms_segments(boxes, scores, None, offs, 0.7, 0)
    b.record()
    torch.cuda.synchronize()
    c = counts.tolist()
    sig = sum(int(keep[o:o + k].sum()) * (i + 1) for i, (o, k) in enumerate(zip(offs, c)))
    print("%.1f us per call, kept per segment %s, checksum %d" % (a.elapsed_time(b) / 30 * 1e3, c, sig))


if __name__ == "__main__":
    run()
