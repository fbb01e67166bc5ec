#!/usr/bin/env python3
"""cProfile of the test-time forward's host side (bench.inference_leg's workload): top functions by cumulative and own
time over 10 forwards; time inside synchronize / .item() / .cpu() is waiting for the device.
    python tools/host_profile_infer.py [N]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda", 0)
    from pet.lib.ops import _hip
    _hip.set_conv_math("bf16x3")
    tr = bench.Trainer(dev)
    images, targets = bench.synthetic_batch(2, 800, 1333, 16, 1234, dev)
    cal, _ = bench.synthetic_batch(2, 800, 1333, 1, 4321, dev)
    bench.calibrate_frozen_affine(tr.model, cal.tensors)
    for _ in range(3):
        tr.step(images, targets)
    r = bench.inference_leg(tr, images, dev, forwards=10)
    print(r)
    model = tr.model
    post = model.Grid_Cascade_RCNN.cls_post_processor
    post.score_thresh = r["score_thresh"]
    model.eval()
    with torch.no_grad():
        for i in range(3):
            model(images.tensors[i % 2:i % 2 + 1])
        torch.cuda.synchronize()
        # device time of one forward: events around it, host made to wait first
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for i in range(10):
            model(images.tensors[i % 2:i % 2 + 1])
        b.record()
        torch.cuda.synchronize()
        print("10 forwards: host+device %.2f ms each, events %.2f ms each" % ((time.perf_counter() - t0) * 100, a.elapsed_time(b) / 10))
        pr = cProfile.Profile()
        pr.enable()
        for i in range(10):
            model(images.tensors[i % 2:i % 2 + 1])
        torch.cuda.synchronize()
        pr.disable()
    st = pstats.Stats(pr)
    st.strip_dirs()
    print("==== by cumulative time (10 forwards)")
    st.sort_stats("cumulative").print_stats(n)
    print("==== by own time (10 forwards)")
    st.sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
