"""Where does the replayed static part differ from the eager one?  (features, objectness, deltas; before / after an
in-place change of an RPN-head weight)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
from bench import Trainer, calibrate_frozen_affine, synthetic_batch
from pet.lib.ops import _hip
_hip.set_conv_math("bf16x3"); _hip.set_deterministic(True)
dev = torch.device("cuda", 0)
tr = Trainer(dev)
images, _ = synthetic_batch(2, 320, 448, 5, 11, dev)
calibrate_frozen_affine(tr.model, images.tensors)
m = tr.model; m.eval()
part = m._static_part()
a = images.tensors[0:1]


def cmp(tag, xs, ys):
    names = ["P%d" % i for i in range(2, 7)] + ["obj%d" % i for i in range(5)] + ["reg%d" % i for i in range(5)]
    bad = [(n, float((x - y).abs().max())) for n, x, y in zip(names, xs, ys) if not torch.equal(x, y)]
    print(tag, "identical" if not bad else bad)


with torch.no_grad():
    e1 = [t.clone() for t in part(a)]
    e2 = [t.clone() for t in part(a)]
    cmp("eager vs eager", e1, e2)
    g1 = [t.clone() for t in m._eval_static(a)]
    cmp("graph (capture) vs eager", g1, e1)
    g2 = [t.clone() for t in m._eval_static(a)]
    cmp("graph (replay) vs eager", g2, e1)
    w = m.RPN.head.conv.weight
    w.mul_(1.5); w.add_(0.01)
    e3 = [t.clone() for t in part(a)]
    e4 = [t.clone() for t in part(a)]
    cmp("changed: eager vs eager", e3, e4)
    g3 = [t.clone() for t in m._eval_static(a)]
    cmp("changed: graph (re-capture) vs eager", g3, e3)
    g4 = [t.clone() for t in m._eval_static(a)]
    cmp("changed: graph (replay) vs eager", g4, e3)
    e5 = [t.clone() for t in part(a)]
    cmp("changed: eager again vs eager", e5, e3)
    print("graphs:", [(k[0], bool(v)) for k, v in m._eval_graphs.items()])


def dets(res):
    r = res[0]
    return r.bbox.clone(), r.get_field("scores").clone(), r.get_field("labels").clone()


def cmpd(tag, x, y):
    print(tag, [bool(a.shape == b.shape and torch.equal(a, b)) for a, b in zip(x, y)],
          [float((a - b).abs().max()) if a.shape == b.shape and a.dtype.is_floating_point else None for a, b in zip(x, y)])


from bench import inference_leg
os.environ["CPM_EVAL_GRAPH"] = "0"
thr = inference_leg(tr, images, dev, forwards=2, rank_cut=80)["score_thresh"]
m.eval()
m.Grid_Cascade_RCNN.cls_post_processor.score_thresh = thr
with torch.no_grad():
    d1, d2 = dets(m(a)), dets(m(a))
    cmpd("full forward eager vs eager", d1, d2)
    os.environ["CPM_EVAL_GRAPH"] = "1"
    d3, d4 = dets(m(a)), dets(m(a))
    cmpd("full forward graph vs eager", d3, d1)
    cmpd("full forward graph replay vs eager", d4, d1)
    os.environ["CPM_EVAL_GRAPH"] = "0"
    d5 = dets(m(a))
    cmpd("full forward eager (3rd) vs eager", d5, d1)
