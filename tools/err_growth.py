#!/usr/bin/env python3
"""Where does the distance between the two conv arithmetics come from at FULL size?  Runs the R-50 body + FPN of the
benchmark model on the benchmark batch in exact-f32 and in bf16x3 and prints, per block output: max |d| / max |ref|,
rms(d) / rms(ref), and the share of elements beyond 1e-3 of the maximum.  Second table: every block fed with the F32
run's input (the arithmetic's own error per block, no propagation)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def stats(a, b):
    d = (a - b).double()
    mx = float(b.abs().max())
    return (float(d.abs().max()) / mx, float(d.pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt()),
            float((d.abs() > 1e-3 * mx).double().mean()))


def main():
    from bench import Trainer, calibrate_frozen_affine, synthetic_batch
    from pet.lib.ops import _hip
    import pet.lib.ops as ops
    dev = torch.device("cuda", 0)
    _hip.set_conv_math("f32")
    tr = Trainer(dev)
    images, _ = synthetic_batch(2, 800, 1333, 16, 1234, dev)
    cal, _ = synthetic_batch(2, 800, 1333, 1, 4321, dev)
    calibrate_frozen_affine(tr.model, cal.tensors)
    body = tr.model.Conv_Body

    def run(mode, feed=None):
        _hip.set_conv_math(mode)
        outs = {}
        with torch.no_grad():
            x = ops.stem_forward(images.tensors, body._stem_weight(), body.bn1.weight, body.bn1.bias, 7, 7, 2, 3,
                                 w=body.conv1.weight) if hasattr(body, "_stem_weight") else None
            outs["stem"] = x
            for li in range(1, 5):
                for bi, blk in enumerate(getattr(body, "layer%d" % li)):
                    name = "layer%d.%d" % (li, bi)
                    xin = x if feed is None else feed[prev_name[name]]
                    x = blk(xin)
                    outs[name] = x
        return outs
    names = ["stem"] + ["layer%d.%d" % (li, bi) for li in range(1, 5) for bi in range(len(getattr(body, "layer%d" % li)))]
    prev_name = {n: names[i - 1] for i, n in enumerate(names) if i}
    ref = run("f32")
    got = run("bf16x3")
    print("%-12s %10s %10s %10s   | own error of the block on the f32 input" % ("output", "max/max", "rms/rms", ">1e-3"))
    own = run("bf16x3", feed=ref)
    for n in names:
        a = stats(got[n], ref[n])
        b = stats(own[n], ref[n]) if n != "stem" else a
        print("%-12s %10.2e %10.2e %10.2e   | %10.2e %10.2e %10.2e" % ((n,) + a + b))


if __name__ == "__main__":
    main()
