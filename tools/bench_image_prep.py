"""Time cpm_image_prep on COCO-shaped inputs (HBM bound: algorithmic bytes = uint8 source read once + fp32 slot written
once) and the host chain it replaces (PIL resize + ToTensor + Normalize + pad, one core)."""
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpm-r-cnn_amd"))
import pet.lib.ops as ops  # noqa: E402

MEAN, STD = [102.9801, 115.9465, 122.7717], [1.0, 1.0, 1.0]


def main():
    rng = np.random.default_rng(0)
    lut = ops.value_table(MEAN, STD, True).cuda()
    batch = torch.empty((2, 3, 800, 1344), device="cuda").contiguous(memory_format=torch.channels_last)
    for (h, w, oh, ow) in [(480, 640, 800, 1066), (427, 640, 800, 1199), (1200, 1600, 800, 1066)]:
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        src = torch.from_numpy(im).cuda()
        for _ in range(5):
            ops.image_prep(src, (oh, ow), False, lut, True, batch[0])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 200
        for _ in range(n):
            ops.image_prep(src, (oh, ow), False, lut, True, batch[0])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        byts = h * w * 3 + 800 * 1344 * 3 * 4
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            t = torch.from_numpy(np.asarray(Image.fromarray(im).resize((ow, oh), Image.BILINEAR)).copy())
            t = t.permute(2, 0, 1).float().div(255)
            t = (t[[2, 1, 0]] * 255).sub_(torch.tensor(MEAN)[:, None, None]).div_(torch.tensor(STD)[:, None, None])
            out = torch.zeros((3, 800, 1344))
            out[:, :oh, :ow] = t
        host_ms = (time.perf_counter() - t0) * 1e3 / reps
        print("%dx%d -> %dx%d (slot 800x1344): device %.1f us  %.0f GB/s of 8000 (%.1f%%) on %.1f MB | host chain "
              "%.1f ms (1 core, %d threads torch)" % (h, w, oh, ow, us, byts / us / 1e3, byts / us / 1e3 / 80,
                                                        byts / 1e6, host_ms, torch.get_num_threads()))


if __name__ == "__main__":
    main()
