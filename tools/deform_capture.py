#!/usr/bin/env python3
"""The fused deformable kernels on the tensors of a real training step (bench.py's X-101-64x4d-FPN-DCN leg): every
layer's backward inputs are captured in one step and the kernels timed on them alone, beside the same call on dense
random gradients of the same size.
    python tools/deform_capture.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    from bench import Trainer, calibrate_frozen_affine, synthetic_batch
    from time_deform import timed
    from pet.lib.ops import _hip as H
    dev = torch.device("cuda", 0)
    tr = Trainer(dev, body="x101dcn")
    images, targets = synthetic_batch(1, 800, 1333, 16, 1234, dev)
    cal, _ = synthetic_batch(1, 800, 1333, 1, 4321, dev)
    calibrate_frozen_affine(tr.model, cal.tensors)
    for _ in range(steps):
        tr.step(images, targets)
    dc = sys.modules["pet.lib.ops.deform_conv"]
    dc._capture = []
    tr.step(images, targets)
    cap, dc._capture = dc._capture, None
    torch.cuda.synchronize()
    L = H.lib()
    s = H.stream()
    for i, (dpre, x, off, w, fa) in enumerate(cap):
        dx = torch.zeros_like(x)
        rnd = torch.randn_like(dpre)
        t_real = timed(lambda: L.cpm_deform_conv_backward_data(H.ptr(dpre), H.ptr(off), H.ptr(w), *fa, H.ptr(dx), s), 5)
        t_rand = timed(lambda: L.cpm_deform_conv_backward_data(H.ptr(rnd), H.ptr(off), H.ptr(w), *fa, H.ptr(dx), s), 5)
        a = dpre.abs()
        print("%2d C=%d %dx%d stride %d: real %.0f us, random %.0f us | dpre zeros %.3f  max %.3g  min>0 %.3g  nan %d inf %d "
              "| off max %s" % (i, fa[3], fa[1], fa[2], fa[7], t_real, t_rand, float((dpre == 0).float().mean()),
                                float(a.max()), float(a[a > 0].min()) if bool((a > 0).any()) else 0.0,
                                int(torch.isnan(dpre).sum()), int(torch.isinf(dpre).sum()),
                                "-" if off is None else "%.3g" % float(off.abs().max())), flush=True)


if __name__ == "__main__":
    main()
