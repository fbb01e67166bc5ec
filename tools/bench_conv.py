#!/usr/bin/env python3
"""Per-layer microbenchmark of the conv kernels at the R-50-FPN CPM R-CNN shapes (bs=2, 800x1344).
Prints TFLOP/s (algorithmic flops) for forward, data gradient and weight gradient of each distinct layer shape.
    python tools/bench_conv.py [--filter NAME] [--iters 10]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))
import torch  # noqa: E402

from pet.lib.ops import conv as ops  # noqa: E402

CL = torch.channels_last
# name, N, C, H, W, K, R, stride, pad, groups, count-per-step (fwd)
LAYERS = [
    ("l1_1x1_64_64", 2, 64, 200, 336, 64, 1, 1, 0, 1, 3),
    ("l1_3x3_64", 2, 64, 200, 336, 64, 3, 1, 1, 1, 3),
    ("l1_1x1_64_256", 2, 64, 200, 336, 256, 1, 1, 0, 1, 4),
    ("l1_1x1_256_64", 2, 256, 200, 336, 64, 1, 1, 0, 1, 2),
    ("l2_1x1s2_256_128", 2, 256, 200, 336, 128, 1, 2, 0, 1, 1),
    ("l2_ds_256_512", 2, 256, 200, 336, 512, 1, 2, 0, 1, 1),
    ("l2_3x3_128", 2, 128, 100, 168, 128, 3, 1, 1, 1, 4),
    ("l2_1x1_128_512", 2, 128, 100, 168, 512, 1, 1, 0, 1, 4),
    ("l2_1x1_512_128", 2, 512, 100, 168, 128, 1, 1, 0, 1, 3),
    ("l3_3x3_256", 2, 256, 50, 84, 256, 3, 1, 1, 1, 6),
    ("l3_1x1_256_1024", 2, 256, 50, 84, 1024, 1, 1, 0, 1, 6),
    ("l3_1x1_1024_256", 2, 1024, 50, 84, 256, 1, 1, 0, 1, 5),
    ("l4_3x3_512", 2, 512, 25, 42, 512, 3, 1, 1, 1, 3),
    ("l4_1x1_512_2048", 2, 512, 25, 42, 2048, 1, 1, 0, 1, 3),
    ("l4_1x1_2048_512", 2, 2048, 25, 42, 512, 1, 1, 0, 1, 2),
    ("fpn_lat_p2", 2, 256, 200, 336, 256, 1, 1, 0, 1, 1),
    ("fpn_out_p2", 2, 256, 200, 336, 256, 3, 1, 1, 1, 2),     # + RPN conv on P2
    ("fpn_out_p3", 2, 256, 100, 168, 256, 3, 1, 1, 1, 2),
    ("fpn_out_p4", 2, 256, 50, 84, 256, 3, 1, 1, 1, 2),
    ("grid_conv0_R64", 64, 256, 14, 14, 576, 3, 2, 1, 1, 3),
    ("grid_conv_R64", 64, 576, 7, 7, 576, 3, 1, 1, 1, 21),
    ("grid_conv_R192", 192, 576, 7, 7, 576, 3, 1, 1, 1, 21),
    ("grid_conv_R105", 105, 576, 7, 7, 576, 3, 1, 1, 1, 0),
    ("grid_conv_R34", 34, 576, 7, 7, 576, 3, 1, 1, 1, 0),
    ("grid_conv_R128", 128, 576, 7, 7, 576, 3, 1, 1, 1, 0),
    ("grid_conv_R160", 160, 576, 7, 7, 576, 3, 1, 1, 1, 0),
    ("grid_conv_R256", 256, 576, 7, 7, 576, 3, 1, 1, 1, 0),
    ("fc6_R1024", 1024, 256, 7, 7, 1024, 7, 1, 0, 1, 2),
    ("fc7_R1024", 1024, 1024, 1, 1, 1024, 1, 1, 0, 1, 2),
    ("iou_fc1_R64", 64, 576, 7, 7, 1024, 7, 1, 0, 1, 1),
    # X-101-64x4d-FPN-DCN at bs=1 (BASELINE config #5): the dense convs around the grouped / deformable 3x3s
    ("x_l1_1x1_256_256", 1, 256, 200, 336, 256, 1, 1, 0, 1, 5),
    ("x_l2_1x1_512_512", 1, 512, 100, 168, 512, 1, 1, 0, 1, 7),
    ("x_l3_1x1_1024_1024", 1, 1024, 50, 84, 1024, 1, 1, 0, 1, 45),
    ("x_l4_1x1_2048_2048", 1, 2048, 25, 42, 2048, 1, 1, 0, 1, 5),
    ("x_l3_offset_1024_18", 1, 1024, 50, 84, 18, 3, 1, 1, 1, 23),
    ("x_l3_offset_1024_20", 1, 1024, 50, 84, 20, 3, 1, 1, 1, 0),      # (the same padded to a multiple of 4 / to an MFMA tile)
    ("x_l3_offset_1024_32", 1, 1024, 50, 84, 32, 3, 1, 1, 1, 0),
    ("x_l2_offset_512_20", 1, 512, 100, 168, 20, 3, 1, 1, 1, 0),
    ("x_l4_offset_2048_20", 1, 2048, 25, 42, 20, 3, 1, 1, 1, 0),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--epi", default="plain", choices=["plain", "res"], help="res: forward with frozen affine + residual "
                    "+ ReLU (bottleneck conv3), data gradient accumulating into an existing tensor")
    ap.add_argument("--math", default="f32", choices=["f32", "bf16x3", "w4"],
                    help="w4 = bf16x3 with the forward weight given as its pre-split image (cpm_split_w4)")
    ap.add_argument("--only", default="", choices=["", "fwd", "dgrad", "wgrad"], help="run one direction only (per-"
                    "direction counter passes: tools/pmc_traffic.sh)")
    a = ap.parse_args()
    from pet.lib.ops import _hip
    use_w4 = a.math == "w4"
    if use_w4:
        a.math = "bf16x3"
    _hip.set_conv_math(a.math)
    tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
    print("%-20s %9s | %8s %7s | %8s %7s | %8s %7s" % ("layer", "GFLOP", "fwd us", "TF/s", "dgrad us", "TF/s",
                                                       "wgrad us", "TF/s"))
    for name, N, C, H, W, K, R, st, pad, g, cnt in LAYERS:
        if a.filter and not any(f in name for f in a.filter.split(",")):
            continue
        x = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=CL)
        w = (torch.randn(K, C // g, R, R, device="cuda") * 0.05).contiguous(memory_format=CL)
        P, Q = ops.out_size(H, R, st, pad), ops.out_size(W, R, st, pad)
        dy = torch.randn(N, K, P, Q, device="cuda").contiguous(memory_format=CL)
        dw = torch.zeros_like(w)
        gf = 2.0 * N * P * Q * K * R * R * (C // g) / 1e9
        err = ""
        fwd = lambda: ops.conv2d_forward(x, w, None, None, None, 0, False, st, pad, 1, g)
        dgr = lambda: ops.conv2d_backward_data(dy, w, (N, C, H, W), st, pad, 1, g)
        if a.epi == "res":
            sc, sh = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda")
            res = torch.randn_like(dy)
            acc = torch.zeros_like(x)
            fwd = lambda: ops.conv2d_forward(x, w, sc, sh, res, 0, True, st, pad, 1, g)
            dgr = lambda: ops.conv2d_backward_data(dy, w, (N, C, H, W), st, pad, 1, g, accumulate_into=acc)
        if use_w4:
            w4 = ops.split_w4(w)
            fwd = lambda: ops.conv2d_forward(x, w, None, None, None, 0, False, st, pad, 1, g, w4=w4)
        if a.math != "f32" and a.epi == "plain" and not a.only:
            y1 = fwd()
            d1 = dgr()
            _hip.set_conv_math("f32")
            y0 = ops.conv2d_forward(x, w, None, None, None, 0, False, st, pad, 1, g)
            d0 = ops.conv2d_backward_data(dy, w, (N, C, H, W), st, pad, 1, g)
            _hip.set_conv_math(a.math)
            err = "  err fwd %.1e dgrad %.1e" % (float((y1 - y0).abs().max() / y0.abs().max()),
                                               float((d1 - d0).abs().max() / d0.abs().max()))
        t_f = timeit(fwd, a.iters) if a.only in ("", "fwd") else 1e9
        t_d = timeit(dgr, a.iters) if a.only in ("", "dgrad") else 1e9
        t_w = timeit(lambda: ops.conv2d_backward_weight(x, dy, w, st, pad, 1, g, out=dw), a.iters) if a.only in ("", "wgrad") else 1e9
        print("%-20s %9.1f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f" % (
            name, gf, t_f * 1e3, gf / t_f, t_d * 1e3, gf / t_d, t_w * 1e3, gf / t_w) + err)
        for k, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
            tot[k][0] += gf * cnt
            tot[k][1] += t * cnt
    for k, (gf, ms) in tot.items():
        if ms:
            print("weighted %-6s: %8.1f GFLOP in %7.2f ms -> %6.1f TF/s" % (k, gf, ms, gf / ms))


if __name__ == "__main__":
    main()
