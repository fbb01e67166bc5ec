#!/usr/bin/env python3
"""How sparse are the gradients that enter the convolutions' backward kernels in the benchmark step?  For every
data-gradient call: the share of 8 x 16 output-pixel patches (the 3x3 patch kernel's workgroup tile) of dy that hold any
nonzero, and the share of nonzero pixels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from bench import Trainer, calibrate_frozen_affine, synthetic_batch
    from pet.lib.ops import _hip
    from pet.lib.ops import conv as C
    dev = torch.device("cuda", 0)
    _hip.set_conv_math("bf16x3")
    tr = Trainer(dev)
    images, targets = synthetic_batch(2, 800, 1333, 16, 1234, dev)
    cal, _ = synthetic_batch(2, 800, 1333, 1, 4321, dev)
    calibrate_frozen_affine(tr.model, cal.tensors)
    for _ in range(3):
        tr.step(images, targets)
    rows = []
    real = C.conv2d_backward_data

    def spy(dy, w, x_shape, *a, **k):
        n, kk, p, q = dy.shape
        if p >= 8 and q >= 16:
            nz = (dy != 0).any(dim=1)                                   # [n, p, q]
            ph, pw = (p + 7) // 8, (q + 15) // 16
            pad = torch.zeros((n, ph * 8, pw * 16), dtype=torch.bool, device=dy.device)
            pad[:, :p, :q] = nz
            tiles = pad.view(n, ph, 8, pw, 16).any(dim=4).any(dim=2)
            rows.append((tuple(dy.shape), tuple(w.shape[2:]), float(nz.float().mean()), float(tiles.float().mean())))
        return real(dy, w, x_shape, *a, **k)
    C.conv2d_backward_data = spy
    tr.step(images, targets)
    torch.cuda.synchronize()
    C.conv2d_backward_data = real
    print("%-28s %-8s %10s %12s" % ("dy shape", "kernel", "nz pixels", "nz patches"))
    for r in rows:
        print("%-28s %-8s %10.3f %12.3f" % (str(r[0]), str(r[1]), r[2], r[3]))


if __name__ == "__main__":
    main()
