"""CPU-only, world_size 2 over gloo: the flat-buffer gradient reducer (pet/utils/parallel.py) and the fused
loss-scalar reduce -- the N > 1 path of bench.py -- plus the flat optimizer's buffer layout."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    from conftest import PKG, ROOT
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pet.utils.optimizer import FlatSGD
        from pet.utils.parallel import FlatGradReducer, reduce_losses
        torch.manual_seed(0)                                    # same weights on every rank
        net = torch.nn.Sequential(torch.nn.Linear(13, 37), torch.nn.ReLU(), torch.nn.Linear(37, 5),
                                  torch.nn.Conv2d(1, 3, 3))
        named = [(k, p, 1 if "bias" in k else 0) for k, p in net.named_parameters()]
        named.reverse()
        opt = FlatSGD(named, [dict(weight_decay=1e-4, lr_scale=1), dict(weight_decay=0.0, lr_scale=2),
                              dict(weight_decay=0.0, lr_scale=1)], 0.9)
        # layout: every tensor is a view into the flat buffers, 256-byte aligned, reverse registration order
        for (_, p, _), b in zip(named, opt.seg_begin.tolist()):
            assert p.data_ptr() == opt.flat_param.data_ptr() + 4 * b and b % 64 == 0
            assert p.grad.data_ptr() == opt.flat_grad.data_ptr() + 4 * b
        red = FlatGradReducer(opt, num_chunks=3, overlap=True)   # overlap silently off on CPU tensors
        assert opt.grad_scale == 1.0 / world
        covered = sorted((b, e) for b, e, _ in red.chunks)
        assert covered[0][0] == 0 and covered[-1][1] == opt.seg_end.tolist()[-1]
        assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
        assert sum(n for _, _, n in red.chunks) == len(named)
        torch.manual_seed(100 + rank)                            # different data per rank
        x = torch.randn(4, 13)
        opt.zero_grad()
        red.begin_step()
        y = net[2](net[1](net[0](x)))
        loss = (y ** 2).mean() + (net[3](torch.randn(2, 1, 6, 6)) ** 2).mean()
        loss.backward()
        local = opt.flat_grad.clone()
        red.finish()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        assert torch.allclose(opt.flat_grad, sum(gathered), rtol=1e-6, atol=1e-7)
        # another cut of the same buffer (bench.py --chunks a,b,c): the first reducer is detached, the new one reduces
        red.close()
        red2 = FlatGradReducer(opt, num_chunks=2, overlap=True)
        assert 1 <= len(red2.chunks) <= 2 and red2.chunks[0][0] == 0 and red2.chunks[-1][1] == opt.seg_end.tolist()[-1]
        opt.flat_grad.copy_(local)
        red2.begin_step()
        red2.finish()
        assert torch.allclose(opt.flat_grad, sum(gathered), rtol=1e-6, atol=1e-7)
        out = reduce_losses({"loss_b": torch.tensor(float(rank + 1)), "loss_a": torch.tensor(10.0 * (rank + 1))})
        assert abs(out["loss_a"] - 15.0) < 1e-6 and abs(out["loss_b"] - 1.5) < 1e-6
        q.put((rank, "ok"))
    except Exception as e:      # surface the failure in the parent
        q.put((rank, "FAIL %r" % (e,)))
        raise
    finally:
        dist.destroy_process_group()


def _bcast_worker(rank, world, port, q):
    import sys
    from conftest import PKG, ROOT
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pet.utils.optimizer import FlatSGD
        from pet.utils.parallel import broadcast_initial_state
        torch.manual_seed(1000 + 17 * rank)                     # every rank initialises DIFFERENT weights
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Linear(13, 37))
        net[0].weight.requires_grad_(False)                     # a frozen tensor outside the flat buffer
        net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)
        named = [(k, p, 1 if "bias" in k else 0) for k, p in net.named_parameters() if p.requires_grad]
        named.reverse()
        opt = FlatSGD(named, [dict(weight_decay=1e-4, lr_scale=1), dict(weight_decay=0.0, lr_scale=2),
                              dict(weight_decay=0.0, lr_scale=1)], 0.9)
        opt.flat_mom.fill_(float(rank + 1))
        opt._steps = 3 * (1 - rank)
        before = [torch.zeros_like(opt.flat_param) for _ in range(world)]
        dist.all_gather(before, opt.flat_param)
        assert not torch.equal(before[0], before[1]), "the test needs different initial weights per rank"
        broadcast_initial_state(net, opt, src=0)
        for t in [opt.flat_param, opt.flat_mom, net[0].weight.data.contiguous(), net[1].running_var,
                  net[1].running_mean]:
            got = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(got, t.contiguous())
            assert torch.equal(got[0], got[1])
        assert torch.equal(opt.flat_param, before[0]) and opt._steps == 3
        assert float(opt.flat_mom[0]) == 1.0
        # parameters still alias the flat buffer after the broadcast
        for (_, p, _), b in zip(named, opt.seg_begin.tolist()):
            assert p.data_ptr() == opt.flat_param.data_ptr() + 4 * b
        q.put((rank, "ok"))
    except Exception as e:
        q.put((rank, "FAIL %r" % (e,)))
        raise
    finally:
        dist.destroy_process_group()


def _run_world2(worker):
    """two spawned ranks over gloo on 127.0.0.1; a rendezvous that fails (the probed port was taken in between, a
    slow spawn) is tried once more on another port -- an assertion inside a worker is not retried"""
    import queue
    for attempt in range(2):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            results = [q.get(timeout=180) for _ in procs]
        except queue.Empty:
            results = []
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if sorted(results) == [(0, "ok"), (1, "ok")]:
            return
        rendezvous = not results or any("AssertionError" not in r[1] and r[1] != "ok" for r in results)
        if attempt == 0 and rendezvous and not any("AssertionError" in r[1] for r in results):
            continue
        assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_broadcast_initial_state_world2():
    """ranks that initialise their heads unseeded must hold rank 0's weights before the first step
    (the reference gets this from DistributedDataParallel's constructor, tools/rcnn/train_net.py:134-136)."""
    _run_world2(_bcast_worker)


def test_flat_reducer_world2():
    _run_world2(_worker)


def test_lr_schedule_matches_reference_formula():
    """warm-up (linear, factor 0.1, 500 it) then x0.1 steps at 120k/160k, bias groups at 2x (lr_scheduler.py:69-127)."""
    from pet.rcnn.core import config
    from pet.utils.lr_scheduler import LearningRateScheduler
    config.reset_cfg()
    config.merge_cfg_from_list(["SOLVER.BASE_LR", 0.02, "SOLVER.STEPS", [120000, 160000], "SOLVER.MAX_ITER", 180000])
    w = torch.nn.Parameter(torch.zeros(3))
    b = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([dict(params=[w], lr=0, lr_scale=1), dict(params=[b], lr=0, lr_scale=2)], momentum=0.9)
    s = LearningRateScheduler(opt, config.cfg.SOLVER, start_iter=0)
    s.step()
    assert abs(s.new_lr - 0.02 * (0.1 * (1 - 1 / 500) + 1 / 500)) < 1e-12
    assert abs(opt.param_groups[1]["lr"] - 2 * s.new_lr) < 1e-12
    s.step(500)
    assert abs(s.new_lr - 0.02) < 1e-12
    s.step(120000)
    assert abs(s.new_lr - 0.002) < 1e-12
    s.step(170000)
    assert abs(s.new_lr - 0.0002) < 1e-12
    config.reset_cfg()
