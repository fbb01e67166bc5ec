#!/usr/bin/env python3
"""torch.profiler over three default training steps: which ATen ops (not this package's kernels) still run in the step.
    python tools/torch_prof_step.py [x101dcn] > gpurun_out/torch_prof.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    import __graft_entry__ as entry
    entry.ensure_built()
    from pet.lib.ops import _hip
    _hip.set_conv_math("bf16x3")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    body = sys.argv[1] if len(sys.argv) > 1 else "resnet"          # "x101dcn": config #5 at batch 1
    bs = 1 if body == "x101dcn" else 2
    tr = bench.Trainer(dev, body=body)
    images, targets = bench.synthetic_batch(bs, 800, 1344, 16, 1234, dev)
    cal, _ = bench.synthetic_batch(bs, 800, 1344, 1, 4321, dev)
    bench.calibrate_frozen_affine(tr.model, cal.tensors)
    for _ in range(6):
        tr.step(images, targets)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(3):
            tr.step(images, targets)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
    # who zero-fills / copies what: the ATen calls with their shapes and Python callers
    seen = {}
    for e in prof.events():
        if e.name in ("aten::zero_", "aten::fill_", "aten::copy_", "aten::add_", "aten::add", "aten::mul", "aten::sum",
                      "aten::cat", "aten::index", "aten::index_select", "aten::contiguous", "aten::clone"):
            shp = str(e.input_shapes[:2])
            st = " <- ".join(f.split("/")[-1] for f in (e.stack or [])[:4] if "torch/" not in f)
            k = (e.name, shp, st)
            a = seen.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
    for k, a in sorted(seen.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%-14s x%-3d %8.1f us  %s  %s" % (k[0], a[0], a[1], k[1], k[2]))
    print(prof.key_averages(group_by_stack_n=4).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60,
                                                      max_src_column_width=90))


if __name__ == "__main__":
    main()
