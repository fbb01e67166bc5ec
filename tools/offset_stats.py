#!/usr/bin/env python3
"""What the offset predictors of bench.py's X-101-64x4d-FPN-DCN leg produce after a few SGD steps: per deformable
layer the share of offsets beyond 1, 2 and 4 pixels (the fused kernels keep samples within ~2 pixels in LDS).
    python tools/offset_stats.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from bench import Trainer, calibrate_frozen_affine, synthetic_batch
    dev = torch.device("cuda", 0)
    tr = Trainer(dev, body="x101dcn")
    images, targets = synthetic_batch(1, 800, 1333, 16, 1234, dev)
    cal, _ = synthetic_batch(1, 800, 1333, 1, 4321, dev)
    calibrate_frozen_affine(tr.model, cal.tensors)
    for _ in range(steps):
        tr.step(images, targets)
    dc = sys.modules["pet.lib.ops.deform_conv"]
    stats = []

    def hook(mod, args, out, name):
        o = out.detach().abs()
        stats.append((name, float(o.mean()), float(o.max()), float((o > 1).float().mean()), float((o > 2).float().mean()),
                      float((o > 4).float().mean())))
    hs = [m.conv_offset.register_forward_hook(lambda mod, a, o, n=n: hook(mod, a, o, n))
          for n, m in tr.model.named_modules() if isinstance(m, dc.DeformConvPack)]
    tr.step(images, targets)
    for h in hs:
        h.remove()
    for s in stats:
        print("%-28s mean %.3g max %.3g  >1: %.3f  >2: %.3f  >4: %.3f" % s)


if __name__ == "__main__":
    main()
