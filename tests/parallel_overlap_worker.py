"""Child rank of tests/test_gpu_00_parallel_overlap.py: one process of a 2-rank data-parallel group on ONE MI355X
(gloo backend, CUDA tensors), exercising pet/utils/parallel.py's overlap path: post-accumulate hooks, the in-place
gradient sinks of the HIP conv / Linear / GroupNorm kernels, chunk all-reduces on a side stream in buffer order.

    python tests/parallel_overlap_worker.py RANK WORLD PORT OUTFILE [BACKEND]

BACKEND "nccl" (= RCCL on ROCm): one GPU per rank (cuda:RANK) -- the configuration the 8-GPU scaling run uses; the
test that passes it only runs on a box with at least two GPUs.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
    dev_index = rank if backend == "nccl" else 0
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    import torch
    import torch.distributed as dist
    import torch.nn as nn
    import pet.lib.ops as ops
    from pet.utils.optimizer import FlatSGD
    from pet.utils.parallel import FlatGradReducer, broadcast_initial_state
    if backend == "nccl":
        torch.cuda.set_device(dev_index)
    dist.init_process_group(backend, rank=rank, world_size=world)
    res = {"rank": rank, "backend": backend}
    try:
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        CL = torch.channels_last

        class Net(nn.Module):
            def __init__(self):
                super().__init__()
                self.c1 = ops.Conv2d(16, 64, 3, 1, 1)
                self.gn = ops.GroupNorm(4, 64)
                self.c2 = ops.Conv2d(64, 64, 3, 2, 1)
                self.fc = ops.Linear(64, 64)             # applied TWICE per step (the use count lives on
                self.out = ops.Linear(64, 5)             # the parameter, the ready hook fires once)

            def forward(self, x):
                x = self.gn(self.c1(x), relu=True)
                x = self.c2(x, relu=True)
                v = x.mean(dim=(2, 3))
                v = self.fc(self.fc(v, relu=True), relu=True)
                return self.out(v)

        torch.manual_seed(1000 + rank)                    # different initial weights per rank on purpose
        net = Net().to(dev).to(memory_format=CL)
        named = [(k, p, 1 if "bias" in k else 0) for k, p in net.named_parameters()]
        named.reverse()
        opt = FlatSGD(named, [dict(weight_decay=1e-4, lr_scale=1), dict(weight_decay=0.0, lr_scale=2),
                              dict(weight_decay=0.0, lr_scale=1)], 0.9)
        # a forward pass under bf16x3 BEFORE the broadcast (bench.py's calibration pass does that)
        # caches a pre-split image of every weight, keyed on the parameter's `_version` -- which a broadcast into the
        # flat buffer does not move.  The broadcast must drop those images: the forward pass after it has to equal
        # the one computed from the float weights, bit for bit, on every rank.
        from pet.lib.ops import _hip
        from pet.lib.ops import conv as conv_ops
        torch.manual_seed(5)
        xq = torch.randn(3, 16, 12, 10, device=dev).contiguous(memory_format=CL)
        _hip.set_conv_math("bf16x3")
        with torch.no_grad():
            net(xq)
        res["images_cached_before_broadcast"] = sum(1 for p_ in net.parameters() if getattr(p_, "_cpm_w4", None) is not None)
        broadcast_initial_state(net, opt, src=0)
        with torch.no_grad():
            y_img = net(xq)
            conv_ops._W4 = False                           # the kernels split the float weights themselves
            y_flt = net(xq)
            conv_ops._W4 = True
        res["stale_image_err"] = float((y_img - y_flt).abs().max())
        res["stale_image_ref_max"] = float(y_flt.abs().max())
        _hip.set_conv_math("f32")
        w0 = [torch.zeros_like(opt.flat_param) for _ in range(world)]
        dist.all_gather(w0, opt.flat_param)
        res["same_start"] = bool(torch.equal(w0[0], w0[1]))
        torch.manual_seed(77 + rank)                      # different batch per rank
        x = torch.randn(4, 16, 12, 10, device=dev).contiguous(memory_format=CL)

        def run():
            y = net(x)
            (y ** 2).mean().backward()

        # local gradient, no reducer attached yet
        opt.zero_grad()
        run()
        torch.cuda.synchronize()
        local = opt.flat_grad.clone()
        # the same network in plain torch on the CPU: the in-place gradient sinks must hold exactly ONE gradient
        import torch.nn.functional as TF
        cp = {k: p.detach().cpu().contiguous().clone().requires_grad_(True) for k, p in net.named_parameters()}
        xc = x.detach().cpu().contiguous()
        h = TF.relu(TF.group_norm(TF.conv2d(xc, cp["c1.weight"], cp["c1.bias"], 1, 1), 4, cp["gn.weight"], cp["gn.bias"]))
        h = TF.relu(TF.conv2d(h, cp["c2.weight"], cp["c2.bias"], 2, 1)).mean(dim=(2, 3))
        h = TF.relu(TF.linear(TF.relu(TF.linear(h, cp["fc.weight"], cp["fc.bias"])), cp["fc.weight"], cp["fc.bias"]))
        (TF.linear(h, cp["out.weight"], cp["out.bias"]) ** 2).mean().backward()
        res["vs_torch"] = {k: float((p.grad.detach().cpu() - cp[k].grad).abs().max() / (cp[k].grad.abs().max() + 1e-30))
                           for k, p in net.named_parameters()}
        red = FlatGradReducer(opt, num_chunks=4, overlap=True)
        res["overlap"] = bool(red.overlap)
        order = []
        real_launch = red._launch

        def recording_launch(ci):
            order.append(ci)
            real_launch(ci)
        red._launch = recording_launch
        fired = []
        names = {id(p): k for k, p in net.named_parameters()}
        for p in net.parameters():
            hook = getattr(p, "_cpm_grad_ready", None)
            if hook is not None:
                def wrapped(q, _h=hook, _n=names[id(p)]):
                    fired.append(_n)
                    _h(q)
                p._cpm_grad_ready = wrapped
            p.register_post_accumulate_grad_hook(lambda q, _n=names[id(p)]: fired.append("autograd:" + _n))
        res["fired_ref"] = fired
        for step in range(2):
            opt.zero_grad()
            red.begin_step()
            run()
            launched_in_backward = list(order)
            red.finish()
            if red.local_sgd:
                opt.step()          # closes the step's ranged updates (learning rate 0 here: the weights stay put)
            torch.cuda.synchronize()
            if step == 0:
                first = opt.flat_grad.clone()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered)
        scale = float(want.abs().max())
        res["err"] = float((opt.flat_grad - want).abs().max() / scale)
        res["err_first"] = float((first - want).abs().max() / scale)
        res["nonzero"] = float(want.abs().max()) > 0
        res["order"] = list(order)                 # (of the two steps above; the list keeps growing below)
        res["chunks"] = len(red.chunks)
        res["launched_in_backward"] = len(launched_in_backward) - len(red.chunks)    # of the 2nd step
        res["ready_fires_per_param_max"] = max(fired.count(i) for i in set(fired)) if fired else 0
        res["autograd_accumulations"] = [f for f in fired if f.startswith("autograd:")]
        res["fc_uses_after"] = int(net.fc.weight._cpm_uses)
        # ---- the optimizer step: behind finish() (default) or chunk by chunk on the reducer's stream behind each chunk's
        # all-reduce (CPM_OVERLAP_SGD=1, FlatSGD.step_range) -- either way the parameters after ONE step are
        # p - lr * (sum of the ranks' gradients / world + wd * p) (first step: the momentum buffer is the gradient)
        res["local_sgd"] = bool(red.local_sgd)
        for g in opt.param_groups:
            g["lr"] = 0.05 * g["lr_scale"]
        p0 = opt.flat_param.clone()
        m0 = opt.flat_mom.clone()
        first_step = opt._steps == 0
        opt.zero_grad()
        red.begin_step()
        run()
        red.finish()
        opt.step()
        torch.cuda.synchronize()
        lr = torch.zeros_like(p0)
        wd = torch.zeros_like(p0)
        live = torch.zeros_like(p0)
        for (b, e), gi in zip(zip(opt.seg_begin.tolist(), opt.seg_end.tolist()), opt._seg_pg):
            lr[b:e] = opt.param_groups[gi]["lr"]
            wd[b:e] = opt.param_groups[gi]["weight_decay"]
            live[b:e] = 1.0
        d = want / world + wd * p0
        expect = p0 - lr * (d if first_step else opt.momentum * m0 + d) * live
        res["sgd_err"] = float((opt.flat_param - expect).abs().max() / float(p0.abs().max()))
        res["sgd_moved"] = float((opt.flat_param - p0).abs().max()) > 0
        allp = [torch.zeros_like(p0) for _ in range(world)]
        dist.all_gather(allp, opt.flat_param)
        res["replicas_equal_after_step"] = float((allp[0] - allp[1]).abs().max())
        res["ok"] = True
    except Exception as e:                                 # surface the failure in the parent
        import traceback
        res["ok"] = False
        res["error"] = "%r\n%s\nfired: %s\nvs_torch: %s" % (e, traceback.format_exc(), res.get("fired_ref"), res.get("vs_torch"))
    finally:
        with open(out, "w") as f:
            json.dump(res, f)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
