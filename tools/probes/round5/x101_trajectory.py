"""Per-step losses, largest |parameter| and offset statistics of one X-101-DCN trainer in one of bench.py's three offset
regimes (zero | bias | trained), as bench.py's side leg sets it up.  CPM_DEFORM_FUSED=0 in the environment: column path."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
from bench import Trainer, calibrate_frozen_affine, synthetic_batch
from pet.lib.ops import _hip
import pet.lib.ops.deform_conv  # noqa
dc = sys.modules["pet.lib.ops.deform_conv"]
_hip.set_conv_math(os.environ.get("PROBE_MATH", "bf16x3"))
dev = torch.device("cuda", 0)
regime = sys.argv[1] if len(sys.argv) > 1 else "bias"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
kw = {"zero": dict(hold_offsets=True), "bias": dict(hold_offsets=True, offset_bias_px=1.5), "trained": dict()}[regime]
tr = Trainer(dev, body="x101dcn", lr_scale=float(os.environ.get("PROBE_LR_SCALE", "1")), **kw)
im, tg = synthetic_batch(1, 800, 1333, 16, 1234, dev)
cal, _ = synthetic_batch(1, 800, 1333, 1, 4321, dev)
calibrate_frozen_affine(tr.model, cal.tensors)
torch.manual_seed(103)
acc = []


def hook(mod, args, out):
    o = out.detach().abs()
    fin = torch.isfinite(o)
    x = args[0].detach()
    acc.append((float(torch.where(fin, o, torch.zeros_like(o)).mean()), float(torch.where(fin, o, torch.zeros_like(o)).max()),
                float((~fin).float().mean()), float((~torch.isfinite(x)).float().mean()), float(x[torch.isfinite(x)].abs().max())))


hs = [m.conv_offset.register_forward_hook(hook) for m in tr.model.modules() if isinstance(m, dc.DeformConvPack)]
names, b, e = tr.optimizer.names, tr.optimizer.seg_begin.tolist(), tr.optimizer.seg_end.tolist()
for s in range(steps):
    del acc[:]
    tr.step(im, tg)
    torch.cuda.synchronize()
    p = tr.optimizer.flat_param
    g = tr.optimizer.flat_grad
    tot = sum(float(v) for v in tr.last_losses.values())
    first_bad = next((i for i, a in enumerate(acc) if a[2] > 0 or a[3] > 0), None)
    print("step %2d loss %10.4f  max|w| %9.3g  max|g| %9.3g  nonfinite w %d  offsets mean %.2f max %.1f  act max %.3g  first bad DCN %s"
          % (s, tot, float(p[torch.isfinite(p)].abs().max()), float(g[torch.isfinite(g)].abs().max()) if bool(torch.isfinite(g).any()) else float("nan"),
             int((~torch.isfinite(p)).sum()), sum(a[0] for a in acc) / len(acc), max(a[1] for a in acc), max(a[4] for a in acc), first_bad), flush=True)
    if int((~torch.isfinite(p)).sum()) > 0:
        bad = [names[i] for i in range(len(names)) if not bool(torch.isfinite(p[b[i]:e[i]]).all())]
        print("   non-finite tensors: %d of %d; first: %s" % (len(bad), len(names), bad[:6]))
        break
