"""Multi-level RoIAlign backward at the bench's shapes (K=1024 7x7 and K=192 14x14), for rocprofv3 --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpm-r-cnn_amd"))
import pet.lib.ops as ops  # noqa: E402

torch.manual_seed(0)
dev = "cuda"
sizes = [(200, 336), (100, 168), (50, 84), (25, 42)]
scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
feats = [torch.randn(2, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
         for h, w in sizes]
for K, ph in ((1024, 7), (192, 14)):
    g = torch.Generator().manual_seed(K)
    wh = torch.rand(K, 2, generator=g) * 368 + 32
    xy = torch.rand(K, 2, generator=g) * (torch.tensor([1333., 800.]) - wh)
    rois = torch.cat([torch.randint(0, 2, (K, 1), generator=g).float(), xy, xy + wh], 1).to(dev)
    y = ops.roi_align_fpn(feats, rois, (ph, ph), scales, 2)
    gy = torch.randn_like(y)
    for it in range(30):
        for f in feats:
            f.grad = None
        ops.roi_align_fpn(feats, rois, (ph, ph), scales, 2).backward(gy)
torch.cuda.synchronize()
print("done")
