import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
from bench import Trainer, calibrate_frozen_affine, synthetic_batch
dev = torch.device("cuda", 0)
tr = Trainer(dev, body="x101dcn", hold_offsets=True)
im, tg = synthetic_batch(1, 800, 1333, 16, 1234, dev)
cal, _ = synthetic_batch(1, 800, 1333, 1, 4321, dev)
calibrate_frozen_affine(tr.model, cal.tensors)
dc = sys.modules["pet.lib.ops.deform_conv"]
for n_steps in (0, 3, 10):
    for _ in range(n_steps):
        tr.step(im, tg)
    rows = []
    def hook(mod, args, out, name):
        x = args[0]
        rows.append((name, int((~torch.isfinite(x)).sum()), float(x.abs().max()), int((~torch.isfinite(out)).sum()), float(out.abs().max()),
                     float(mod.weight.abs().max()), float(mod.bias.abs().max())))
    hs = [m.conv_offset.register_forward_hook(lambda mod, a, o, n=n: hook(mod, a, o, n)) for n, m in tr.model.named_modules() if isinstance(m, dc.DeformConvPack)]
    tr.step(im, tg)
    torch.cuda.synchronize()
    for h in hs: h.remove()
    print("after", n_steps, "more steps; losses", {k: round(float(v), 4) for k, v in tr.last_losses.items()})
    for r in rows[:6] + rows[-3:]:
        print("  %-28s x nonfinite %d max %.3g | off nonfinite %d max %.3g | w %.3g b %.3g" % r)
