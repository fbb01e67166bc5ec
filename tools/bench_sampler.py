#!/usr/bin/env python3
"""cpm_sample_pos_neg on the RPN's and the RoI head's shapes vs the sort-based torch formulation it replaced."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

import pet.lib.ops as ops  # noqa: E402


def torch_formulation(labels, img, n_img, batch, frac):
    """One sort over random keys inside (image, class) buckets + rank-below-quota (the previous fused sampler)."""
    R, dev = labels.numel(), labels.device
    cls = torch.where(labels >= 1, 0, torch.where(labels == 0, 1, 2))
    bucket = img.long() * 3 + cls
    key = bucket.to(torch.float32) + torch.rand(R, device=dev) * 0.998
    order = torch.argsort(key)
    b_sorted = bucket[order]
    bounds = torch.searchsorted(b_sorted, torch.arange(3 * n_img + 1, device=dev))
    starts, c = bounds[:-1], (bounds[1:] - bounds[:-1]).view(n_img, 3)
    n_pos = c[:, 0].clamp(max=int(batch * frac))
    n_neg = torch.minimum(c[:, 1], batch - n_pos)
    quota = torch.stack([n_pos, n_neg, torch.zeros_like(n_pos)], dim=1).view(-1)
    rank = torch.arange(R, device=dev) - starts[b_sorted]
    take = torch.empty(R, dtype=torch.bool, device=dev)
    take[order] = rank < quota[b_sorted]
    return take & (cls == 0), take & (cls == 1)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, counts, batch, frac, dtype in (("rpn  2 x 268569 anchors", [268569, 268569], 256, 0.5, torch.float32),
                                         ("head 2 x ~1016 proposals", [1011, 1021], 512, 0.25, torch.int64)):
    R = sum(counts)
    g = torch.Generator().manual_seed(0)
    lab = torch.where(torch.rand(R, generator=g) < 0.3, -1, 0)
    lab[torch.randint(0, R, (120,), generator=g)] = 1
    lab = lab.to(dtype).cuda()
    img = torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(counts)).cuda()
    t_new = timeit(lambda: ops.sample_pos_neg(lab, counts, batch, frac))
    t_old = timeit(lambda: torch_formulation(lab, img, len(counts), batch, frac))
    print("%-26s cpm_sample_pos_neg %7.1f us   sort-based torch formulation %7.1f us" % (name, t_new, t_old))
