"""layer1.0 of the X-101 body alone, repeated: conv1's output against torch per repetition."""
import os, sys, torch
import torch.nn.functional as TF
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
from bench import Trainer, calibrate_frozen_affine, synthetic_batch
from pet.lib.ops import _hip
_hip.set_conv_math("bf16x3")
dev = torch.device("cuda", 0)
tr = Trainer(dev, body="x101dcn", hold_offsets=True)
cal, _ = synthetic_batch(1, 800, 1333, 1, 4321, dev)
calibrate_frozen_affine(tr.model, cal.tensors)
blk = tr.model.Conv_Body.layer1[0]
x = torch.randn(1, 64, 200, 336, device=dev).contiguous(memory_format=torch.channels_last)
w = blk.conv1.weight.detach().float()
s, b = blk.bn1.weight.detach(), blk.bn1.bias.detach()
want = torch.relu(TF.conv2d(x.cpu().contiguous(), w.cpu().contiguous()) * s.cpu().view(1, -1, 1, 1) + b.cpu().view(1, -1, 1, 1))
prev = None
for rep in range(4):
    got = {}
    h = blk.conv1.register_forward_hook(lambda m, a, o: got.__setitem__("c1", o.detach().clone()))
    h2 = blk.register_forward_hook(lambda m, a, o: got.__setitem__("blk", o.detach().clone()))
    with torch.no_grad():
        blk(x)
    torch.cuda.synchronize()
    h.remove(); h2.remove()
    e = float((got["c1"].cpu() - want).abs().max() / want.abs().max())
    same = None if prev is None else (bool(torch.equal(prev["c1"], got["c1"])), bool(torch.equal(prev["blk"], got["blk"])))
    print("rep %d: conv1 vs torch %.3g; equal to previous rep (conv1, block): %s" % (rep, e, same))
    prev = got
