"""Which parameters differ after k steps with the optimizer update beside the next forward pass vs behind it?
(deterministic reductions, fixed seeds: the two runs are bit-comparable)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
DET = os.environ.get("PROBE_DET", "1") == "1"
if DET:
    os.environ["CPM_DETERMINISTIC"] = "1"
from bench import Trainer, calibrate_frozen_affine, synthetic_batch
from pet.lib.ops import _hip
_hip.set_conv_math("bf16x3")
_hip.set_deterministic(DET)
dev = torch.device("cuda", 0)
body = sys.argv[1] if len(sys.argv) > 1 else "x101dcn"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def run(overlap):
    torch.manual_seed(0)
    tr = Trainer(dev, body=body, hold_offsets=(body == "x101dcn")) if body != "resnet" else Trainer(dev)
    tr.optimizer.overlap_next_forward = overlap
    bs = 1 if body == "x101dcn" else 2
    im, tg = synthetic_batch(bs, 800, 1333, 16, 1234, dev)
    cal, _ = synthetic_batch(bs, 800, 1333, 1, 4321, dev)
    calibrate_frozen_affine(tr.model, cal.tensors)
    torch.manual_seed(5)
    snaps = []
    for _ in range(steps):
        tr.step(im, tg)
        torch.cuda.synchronize()
        snaps.append((tr.optimizer.flat_param.clone(), {k: float(v) for k, v in tr.last_losses.items()}))
    names, begins, ends = tr.optimizer.names, tr.optimizer.seg_begin.tolist(), tr.optimizer.seg_end.tolist()
    return snaps, names, begins, ends


order = sys.argv[3] if len(sys.argv) > 3 else "off,on,off"
runs = []
for o in order.split(","):
    r = run(o == "on")
    runs.append((o, r[0]))
    names, begins, ends = r[1], r[2], r[3]
    print("run %-3s:" % o, " | ".join("obj %.3f cls %.3f grid1 %.4f" % (l["loss_objectness"], l["loss_classifier"], l["loss_grid_1"]) for _, l in r[0]))
a, b, a2 = runs[0][1], runs[1][1], runs[-1][1]
for s in range(steps):
    same_ref = bool(torch.equal(a[s][0], a2[s][0]))
    d = (a[s][0] - b[s][0]).abs()
    bad = [(names[i], float(d[begins[i]:ends[i]].max()), float(a[s][0][begins[i]:ends[i]].abs().max()))
           for i in range(len(names)) if float(d[begins[i]:ends[i]].max()) > 0]
    print("step %d: off-vs-off identical: %s; tensors that differ off-vs-on: %d of %d" % (s, same_ref, len(bad), len(names)))
    print("   losses off:", {k: round(v, 4) for k, v in a[s][1].items()})
    print("   losses on :", {k: round(v, 4) for k, v in b[s][1].items()})
    for n, e, m in bad[:12]:
        print("      %-60s max diff %.3g (max |w| %.3g)" % (n, e, m))
