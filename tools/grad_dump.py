#!/usr/bin/env python3
"""One forward + backward of the default training batch (ordered reductions); writes every parameter gradient's L2 norm
and a fixed random projection to OUT.npz -- run it under different CPM_* switches and compare:
    python tools/grad_dump.py OUT ; python tools/grad_dump.py --compare A.npz B.npz"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import numpy as np  # noqa: E402


def compare(a, b):
    A, B = np.load(a), np.load(b)
    worst = []
    for k in A.files:
        if k.endswith("/norm"):
            na, nb = float(A[k]), float(B[k])
            pa, pb = A[k[:-5] + "/proj"], B[k[:-5] + "/proj"]
            worst.append((abs(na - nb) / (abs(nb) + 1e-30), float(np.abs(pa - pb).max() / (np.abs(pb).max() + 1e-30)), k[:-5]))
    worst.sort(reverse=True)
    print("losses:", {k[5:]: (float(A[k]), float(B[k])) for k in A.files if k.startswith("loss/")})
    print("worst relative norm differences:", [(round(w[0], 7), w[2]) for w in worst[:5]])
    worst.sort(key=lambda w: -w[1])
    print("worst projection differences  :", [(round(w[1], 7), w[2]) for w in worst[:5]])


def main():
    if sys.argv[1] == "--compare":
        return compare(sys.argv[2], sys.argv[3])
    import torch
    import bench
    import __graft_entry__ as entry
    entry.ensure_built()
    from pet.lib.ops import _hip
    _hip.set_conv_math("bf16x3")
    _hip.set_deterministic(True)
    dev = torch.device("cuda", 0)
    tr = bench.Trainer(dev)
    tr.optimizer.clear_grads_in_step = False
    images, targets = bench.synthetic_batch(2, 800, 1333, 16, 1234, dev)
    cal, _ = bench.synthetic_batch(2, 800, 1333, 1, 4321, dev)
    bench.calibrate_frozen_affine(tr.model, cal.tensors)
    torch.manual_seed(7)
    tr.scheduler.step()
    tr.optimizer.zero_grad()
    tr.reducer.begin_step()
    out = tr.model(images, targets)
    bench.backward_losses_fn(out["losses"])
    torch.cuda.synchronize()
    res = {"loss/" + k: float(v.detach()) for k, v in out["losses"].items()}
    g = torch.Generator(device="cpu").manual_seed(3)
    for name, p in tr.model.named_parameters():
        if p.requires_grad and p.grad is not None:
            gr = p.grad.detach().float().reshape(-1).cpu()
            idx = torch.randint(0, gr.numel(), (64,), generator=g)
            res[name + "/norm"] = float(gr.norm())
            res[name + "/proj"] = gr[idx].numpy()
    np.savez(sys.argv[1], **res)


if __name__ == "__main__":
    main()
