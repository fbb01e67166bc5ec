#!/usr/bin/env python3
"""Where the step's small framework launches come from: one training step under torch.profiler with Python stacks, the
copy / fill / cat / elementwise ops grouped by their innermost frame inside this repository.
    python tools/copy_sources.py"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    from pet.lib.ops import _hip
    _hip.set_conv_math("bf16x3")
    tr = bench.Trainer(dev)
    images, targets = bench.synthetic_batch(2, 800, 1333, 16, 1234, dev)
    cal, _ = bench.synthetic_batch(2, 800, 1333, 1, 4321, dev)
    bench.calibrate_frozen_affine(tr.model, cal.tensors)
    for _ in range(4):
        tr.step(images, targets)
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    groups = collections.Counter()
    want = ("copy_", "fill_", "zero_", "cat", "_to_copy", "clone", "zeros", "ones", "full", "mul", "add", "div", "sub",
            "sum", "index", "index_select", "arange", "stack", "where", "zeros_like", "ones_like", "eq", "ge", "lt",
            "bitwise_and", "masked_fill_", "select", "slice", "nonzero", "empty_strided", "_local_scalar_dense")
    skip = ("select", "slice")

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = func.__name__.split(".")[0]
            if name in want and name not in skip:
                frame = "?"
                for fr in reversed(traceback.extract_stack()):
                    if "/pet/" in fr.filename or fr.filename.endswith("bench.py"):
                        frame = "%s:%d %s" % (fr.filename.split("cpm-r-cnn_amd/")[-1], fr.lineno, fr.name)
                        break
                if name in ("clone", "zeros", "zeros_like", "add"):
                    t = args[0]
                    if hasattr(t, "shape"):
                        frame += "  %s strides %s" % (tuple(t.shape), tuple(t.stride()))
                    else:
                        frame += "  %s" % (t,)
                    # the caller of the conversion
                    st = [fr for fr in traceback.extract_stack() if "/pet/" in fr.filename]
                    if len(st) >= 2:
                        frame += "  <- %s:%d" % (st[-2].filename.split("/pet/")[-1], st[-2].lineno)
                groups[(name, frame)] += 1
            return func(*args, **(kwargs or {}))

    from pet.utils.parallel import backward_losses
    tr.optimizer.zero_grad()
    tr.reducer.begin_step()
    with Spy():
        out = tr.model(images, targets)
    torch.cuda.synchronize()
    print("---- forward")
    for (name, frame), n in sorted(groups.items(), key=lambda kv: (-kv[1], kv[0])):
        print("%3d  %-22s %s" % (n, name, frame))
    groups.clear()
    with torch.autograd.set_multithreading_enabled(False), Spy():          # the backward pass on this thread
        backward_losses(out["losses"])
    torch.cuda.synchronize()
    print("---- backward")
    for (name, frame), n in sorted(groups.items(), key=lambda kv: (-kv[1], kv[0])):
        print("%3d  %-22s %s" % (n, name, frame))


if __name__ == "__main__":
    main()
