"""Run the X-101 body's forward several times on the same input and weights and report the first module whose output
changes between repetitions (a cross-stream ordering / lifetime bug shows up as such a change)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
from bench import Trainer, calibrate_frozen_affine, synthetic_batch
from pet.lib.ops import _hip
_hip.set_conv_math("bf16x3")
dev = torch.device("cuda", 0)
tr = Trainer(dev, body="x101dcn", hold_offsets=True)
im, tg = synthetic_batch(1, 800, 1333, 16, 1234, dev)
cal, _ = synthetic_batch(1, 800, 1333, 1, 4321, dev)
calibrate_frozen_affine(tr.model, cal.tensors)
body = tr.model.Conv_Body
names = [n for n, m in body.named_modules() if (n.count(".") == 1 and n.startswith("layer")) or n.startswith("layer1.0.")]
names = [n for n in names if not n.endswith(".relu") and "bn" not in n and not n.endswith("downsample")]
mods = dict(body.named_modules())
reps = []
for rep in range(4):
    outs = {}
    hs = [mods[n].register_forward_hook(lambda m, a, o, n=n: outs.__setitem__(n, o.detach().clone())) for n in names]
    hs.append(mods["layer1.0"].register_forward_pre_hook(lambda m, a: outs.__setitem__("layer1.0.INPUT", a[0].detach().clone())))
    grad = rep >= 2                      # repetitions 2, 3 with autograd on (the training forward)
    with torch.set_grad_enabled(grad):
        c = body(im.tensors)
    torch.cuda.synchronize()
    for h in hs:
        h.remove()
    reps.append(outs)
names = ["layer1.0.INPUT"] + names
for rep in range(1, 4):
    first = None
    for n in names:
        a, b = reps[0][n], reps[rep][n]
        d = float((a - b).abs().max())
        if d > 0 and first is None:
            first = (n, d, float(a.abs().max()))
        if n.startswith("layer1.0"):
            print("      %s differs by %.3g (max %.3g)" % (n, d, float(a.abs().max())))
    print("repetition %d vs 0 (grad %s): %s" % (rep, rep >= 2, "identical" if first is None else "first difference at %s: %.3g (max %.3g)" % first))
