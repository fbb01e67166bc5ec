"""Directional finite-difference check of the whole training step's gradient on X-101-DCN (f32 conv arithmetic, ordered
reductions, sampler re-seeded before every forward): g.d from one backward pass against (L(w + e d) - L(w - e d)) / 2e for
directions d restricted to groups of parameters.  Run for the `zero` and `bias` offset regimes: the ratio should be the
same (~1) in both."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
os.environ["CPM_DETERMINISTIC"] = "1"
import bench
from bench import Trainer, calibrate_frozen_affine, synthetic_batch
from pet.lib.ops import _hip
_hip.set_conv_math("f32")
_hip.set_deterministic(True)
dev = torch.device("cuda", 0)
regime = sys.argv[1] if len(sys.argv) > 1 else "bias"
kw = {"zero": dict(hold_offsets=True), "bias": dict(hold_offsets=True, offset_bias_px=1.5), "trained": dict()}[regime]
tr = Trainer(dev, body="x101dcn", **kw)
im, tg = synthetic_batch(1, 800, 1333, 16, 1234, dev)
cal, _ = synthetic_batch(1, 800, 1333, 1, 4321, dev)
calibrate_frozen_affine(tr.model, cal.tensors)
opt = tr.optimizer
names, b, e = opt.names, opt.seg_begin.tolist(), opt.seg_end.tolist()


# (the RPN's two terms only: their anchor sample does not depend on the parameters, so the objective is smooth in them;
#  the RoI heads' terms jump whenever a proposal changes)
KEEP = ("loss_objectness", "loss_rpn_box_reg") if os.environ.get("PROBE_TERMS", "rpn") == "rpn" else None


def loss(backward=False):
    torch.manual_seed(77)
    if backward:
        opt.zero_grad()
        tr.reducer.begin_step()
        out = tr.model(im, tg)
        tr.reducer.mark_backward_begin()
        bench.backward_losses_fn({k: v for k, v in out["losses"].items() if KEEP is None or k in KEEP})
        tr.reducer.finish()
    else:
        with torch.no_grad():
            out = tr.model(im, tg)
    torch.cuda.synchronize()
    return {k: float(v.detach().double()) for k, v in out["losses"].items() if KEEP is None or k in KEEP}


base = loss(True)
g = opt.flat_grad.clone()
print(regime, "losses", {k: round(v, 5) for k, v in base.items()}, "total", sum(base.values()))
again = loss(False)
print("  forward repeat delta", sum(again.values()) - sum(base.values()))
groups = [("dcn conv2.weight (layer2-4)", lambda n: "Conv_Body" in n and ".conv2.weight" in n and "layer1" not in n),
          ("body conv1.weight", lambda n: "Conv_Body" in n and ".conv1.weight" in n),
          ("body conv3.weight", lambda n: "Conv_Body" in n and ".conv3.weight" in n),
          ("body downsample", lambda n: "Conv_Body" in n and "downsample" in n),
          ("layer2 all", lambda n: "Conv_Body" in n and "layer2" in n and "conv_offset" not in n),
          ("layer3 all", lambda n: "Conv_Body" in n and "layer3" in n and "conv_offset" not in n),
          ("layer4 all", lambda n: "Conv_Body" in n and "layer4" in n and "conv_offset" not in n),
          ("conv_offset.bias", lambda n: "conv_offset.bias" in n),
          ("conv_offset.weight", lambda n: "conv_offset.weight" in n),
          ("FPN", lambda n: "Conv_Body_FPN" in n or "fpn" in n.lower()),
          ("RPN", lambda n: "RPN" in n),
          ("heads", lambda n: "Grid_Cascade_RCNN" in n)]
gen = torch.Generator(device="cpu").manual_seed(3)
for title, pick in groups:
    idx = [i for i, n in enumerate(names) if pick(n)]
    if not idx:
        print("  %-30s no tensors" % title)
        continue
    d = torch.zeros_like(opt.flat_param)
    for i in idx:
        w = opt.flat_param[b[i]:e[i]]
        s = float(w.std()) if w.numel() > 1 and float(w.std()) > 0 else 1.0
        if "conv_offset" in names[i]:
            s = 0.05
        d[b[i]:e[i]] = (torch.randn(e[i] - b[i], generator=gen) * s).to(dev)
    gd = float((g.double() * d.double()).sum())
    for eps in (1e-4, 3e-5, 1e-5):
        opt.flat_param.add_(d, alpha=eps)
        lp = sum(loss().values())
        opt.flat_param.add_(d, alpha=-2 * eps)
        lm = sum(loss().values())
        opt.flat_param.add_(d, alpha=eps)
        fd = (lp - lm) / (2 * eps)
        print("  %-30s tensors %3d  eps %.0e  g.d %+12.5f  fd %+12.5f  ratio %.4f" % (title, len(idx), eps, gd, fd, fd / gd if gd else float("nan")), flush=True)
