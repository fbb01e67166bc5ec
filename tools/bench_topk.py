"""cpm_topk_rows vs torch.topk at the RPN's five FPN level sizes (2 images, k = min(2000, n))."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpm-r-cnn_amd"))
import pet.lib.ops as ops  # noqa: E402


def timed(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot = [0.0, 0.0]
for n in (201600, 50400, 12600, 3150, 819):
    k = min(2000, n)
    for name, s in (("init", torch.sigmoid(torch.randn(2, n, device="cuda") * 0.02)),
                    ("spread", torch.sigmoid(torch.randn(2, n, device="cuda") * 4 - 4))):
        a = timed(lambda: ops.topk_rows(s, k))
        b = timed(lambda: s.topk(k, dim=1, sorted=True))
        print("n=%6d k=%4d %-6s  cpm_topk_rows %7.1f us   torch.topk %7.1f us" % (n, k, name, a, b))
        if name == "init":
            tot[0] += a
            tot[1] += b
print("all five levels: %.1f us vs %.1f us" % tuple(tot))
