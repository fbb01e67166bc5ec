#!/usr/bin/env python3
"""cProfile of the training step's host side (the Python that issues the launches): top functions by own time and
by cumulative time over 10 steps.  The device runs asynchronously; time inside .item() / synchronize is waiting.
    python tools/host_profile.py [N]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 45
    dev = torch.device("cuda", 0)
    from pet.lib.ops import _hip
    _hip.set_conv_math("bf16x3")
    tr = bench.Trainer(dev)
    images, targets = bench.synthetic_batch(2, 800, 1333, 16, 1234, dev)
    cal, _ = bench.synthetic_batch(2, 800, 1333, 1, 4321, dev)
    bench.calibrate_frozen_affine(tr.model, cal.tensors)
    for _ in range(5):
        tr.step(images, targets)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        tr.step(images, targets)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.strip_dirs()
    print("==== by own time (10 steps)")
    st.sort_stats("tottime").print_stats(n)
    print("==== by cumulative time (10 steps)")
    st.sort_stats("cumulative").print_stats(n)


if __name__ == "__main__":
    main()
