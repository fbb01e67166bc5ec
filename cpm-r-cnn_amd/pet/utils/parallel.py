"""Data parallelism over RCCL/xGMI: one process per GPU, replicated model, flat-buffer gradient all-reduce.

Counterpart of the DistributedDataParallel wrap in tools/rcnn/train_net.py:134-136 and of the per-loss
all_reduce in pet/utils/logger.py:52-53, re-designed for the flat gradient buffer of pet.utils.optimizer:
  * the 153.6 M-element gradient lives in one buffer ordered so that backward completes it front to back;
  * it is cut into a few large chunks (default 8 x ~77 MB: xGMI is point-to-point, large messages amortise
    the ring's per-link latency); a chunk's all-reduce is launched on a side stream as soon as the autograd
    hooks have seen every tensor in it, overlapping the remaining backward;
  * the 1/world scaling is folded into the SGD kernel (grad_scale) instead of a separate division pass;
  * optionally (CPM_OVERLAP_SGD=1) the SGD update of a chunk follows its all-reduce on the same side stream
    (FlatSGD.step_range), beside the rest of the backward pass instead of behind it;
  * the ~9 loss scalars are reduced as ONE tensor for logging.
Works with any torch.distributed backend (`nccl` == RCCL on ROCm; `gloo` in the CPU tests)."""
import os

import torch
import torch.distributed as dist

from pet.lib.ops.conv import wgrad_stream


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class FlatGradReducer(object):
    def __init__(self, optimizer, num_chunks=8, overlap=True):
        self.opt = optimizer
        self.world = world()
        self.opt.grad_scale = 1.0 / self.world
        self.flat = optimizer.flat_grad
        total = self.flat.numel()
        # chunk boundaries on tensor boundaries
        begins = optimizer.seg_begin.tolist()
        ends = optimizer.seg_end.tolist()
        target = (total + num_chunks - 1) // num_chunks
        self.chunks, start, nseg = [], 0, 0
        self.seg_chunk = []
        for b, e in zip(begins, ends):
            self.seg_chunk.append(len(self.chunks))
            nseg += 1
            if e - start >= target or e == ends[-1]:
                self.chunks.append((start, e, nseg))
                start, nseg = e, 0
        self.overlap = overlap and self.world > 1 and self.flat.is_cuda
        # the chunk's SGD update behind its gradients (and all-reduce), beside the backward pass.  Opt-in
        # (CPM_OVERLAP_SGD=1): on one MI355X it changes nothing -- 20.9 vs 21.1 ms/step alternating on one box; the SGD
        # kernel's 3.7 GB of traffic takes from the backward kernels what it saves behind them -- and with RCCL it
        # has not run yet.
        self.local_sgd = overlap and self.flat.is_cuda and os.environ.get("CPM_OVERLAP_SGD", "0") != "0"
        self.stream = torch.cuda.Stream(device=self.flat.device) if (self.overlap or self.local_sgd) else None
        self._pending = None
        self._done = set()
        self._handles = []
        self._hook_handles, self._hooked = [], []
        # timing (bench.py at N > 1; off in training runs): events on the communication stream around every chunk's
        # collective and on the compute stream where the backward pass starts and ends -- comm_stats()
        self.timing = False
        self._ev = None
        if self.overlap or self.local_sgd:
            params = [p for g in optimizer.param_groups for p in g["params"]]
            by_id = {id(p): p for p in params}
            order = []
            for g in optimizer.param_groups:
                order.extend(g["params"])
            # map each param to its chunk through its offset in the flat buffer
            base = self.flat.data_ptr()
            for p in by_id.values():
                off = (p.grad.data_ptr() - base) // 4
                ci = next(i for i, (b, e, _) in enumerate(self.chunks) if b <= off < e)
                hook = self._make_hook(ci, p)
                self._hook_handles.append(p.register_post_accumulate_grad_hook(hook))
                if hasattr(p, "_cpm_grad_sink"):
                    p._cpm_grad_ready = hook        # conv weights bypass autograd's accumulation (pet.lib.ops.conv)
                    self._hooked.append((p, hook))

    def close(self):
        """Detach from the parameters (a trainer that re-cuts its gradient buffer into another number of chunks --
        bench.py's --chunks sweep -- builds a new reducer): this one's hooks are removed and it ignores late calls."""
        for h in self._hook_handles:
            h.remove()
        for p, hook in self._hooked:
            if getattr(p, "_cpm_grad_ready", None) is hook:
                del p._cpm_grad_ready
        self._hook_handles, self._hooked = [], []
        self._pending = None

    def _make_hook(self, ci, p):
        """A parameter's gradient is complete either when autograd has accumulated it (post-accumulate hook) or when
        the last in-place use of the step has been queued by a HIP backward kernel (`_cpm_grad_ready`, called by
        pet.lib.ops.conv when the parameter's use count returns to 0).  Both routes can fire for one parameter in one
        step: torch runs the post-accumulate hooks of a leaf even when the Function handed it no gradient (the sink
        route returns None), and for a weight used twice it does so after the FIRST of the two backward calls.  So: an
        event counts only when no in-place use is outstanding, and only once per step."""
        def hook(_=None):
            if self._pending is None or id(p) in self._done or getattr(p, "_cpm_uses", 0) > 0:
                return
            self._done.add(id(p))
            assert self._pending[ci] > 0, "gradient-ready hook fired more often than chunk %d has tensors" % ci
            self._pending[ci] -= 1
            # chunks are reduced strictly in buffer order on every rank -- a collective must be issued in the same
            # order everywhere, and the order in which chunks BECOME ready may differ between ranks (a rank whose
            # batch leaves a head without RoIs never completes that head's chunk until finish())
            if self.flat.is_cuda and torch.cuda.is_current_stream_capturing():
                return                      # (a hipGraph capture of the backward pass: finish() launches instead)
            while self._next < len(self.chunks) and self._pending[self._next] == 0:
                self._launch(self._next)
                self._next += 1
        return hook

    def begin_step(self):
        self._pending = [n for (_, _, n) in self.chunks]
        self._done = set()
        self._next = 0
        self._handles = []
        self._ev = None
        if self.timing and self.stream is not None:
            E = lambda: torch.cuda.Event(enable_timing=True)
            self._ev = {"chunks": [None] * len(self.chunks), "bwd_begin": E(), "bwd_end": E(), "comm_end": E()}

    def mark_backward_begin(self):
        """timing runs: the trainer calls this right before it starts the backward pass"""
        if self._ev is not None:
            self._ev["bwd_begin"].record(torch.cuda.current_stream(self.flat.device))

    def comm_stats(self):
        """after a synchronize: {backward_ms, per_chunk_ms[], comm_busy_ms, exposed_ms} of the LAST step --
        per_chunk_ms: each chunk's collective on the communication stream (plus its SGD update under CPM_OVERLAP_SGD);
        exposed_ms: how long the compute stream waited for the communication stream behind the end of its backward
        pass (0 = the collectives were hidden completely)"""
        ev = self._ev
        if ev is None or any(c is None for c in ev["chunks"]):
            return None
        per = [a.elapsed_time(b) for a, b in ev["chunks"]]
        return {"backward_ms": round(ev["bwd_begin"].elapsed_time(ev["bwd_end"]), 3),
                "per_chunk_ms": [round(x, 3) for x in per], "comm_busy_ms": round(sum(per), 3),
                "first_chunk_start_ms_into_backward": round(ev["bwd_begin"].elapsed_time(ev["chunks"][0][0]), 3),
                "exposed_ms": round(max(0.0, ev["bwd_end"].elapsed_time(ev["comm_end"])), 3)}

    def _launch(self, ci):
        b, e, _ = self.chunks[ci]
        self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
        side = wgrad_stream(self.flat.device)       # weight gradients are queued on a second stream (pet.lib.ops.conv)
        if side is not None:
            self.stream.wait_stream(side)
        with torch.cuda.stream(self.stream):
            if self._ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self._ev["chunks"][ci] = (e0, e1)
                e0.record(self.stream)
            if self.world > 1:
                dist.all_reduce(self.flat[b:e], op=dist.ReduceOp.SUM)
            if self.local_sgd:
                # (chunks end on tensor ends, tensors start on 64-element boundaries: the gap belongs to nobody)
                self.opt.step_range((b + 63) // 64 * 64, (e + 63) // 64 * 64)
            if self._ev is not None:
                self._ev["chunks"][ci][1].record(self.stream)

    def finish(self):
        """Call after backward: reduce whatever is left and make the compute stream wait for the reductions."""
        if self.world == 1 and not self.local_sgd:
            return
        if self.overlap or self.local_sgd:
            if self._ev is not None:
                self._ev["bwd_end"].record(torch.cuda.current_stream(self.flat.device))
            while self._next < len(self.chunks):   # incl. chunks with tensors that got no gradient this step
                self._launch(self._next)
                self._next += 1
            if self._ev is not None:
                self._ev["comm_end"].record(self.stream)
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        else:
            for b, e, _ in self.chunks:
                dist.all_reduce(self.flat[b:e], op=dist.ReduceOp.SUM)


def broadcast_initial_state(model, optimizer, src=0):
    """Make every rank start from rank `src`'s state -- what DistributedDataParallel's constructor does in the
    reference (tools/rcnn/train_net.py:134-136: it broadcasts module parameters and buffers).  FPN / RPN / head weights
    are randomly initialised per process (nn.init.* in FPN.py, rpn.py, outputs.py); without this each rank would apply
    the averaged gradient to different weights.  ONE broadcast covers every trainable tensor (the flat parameter
    buffer), one the momentum buffer; frozen parameters and buffers (frozen stem / layer1, AffineChannel2d) follow
    tensor by tensor -- they are few and small."""
    if world() == 1:
        return
    dist.broadcast(optimizer.flat_param, src=src)
    dist.broadcast(optimizer.flat_mom, src=src)
    steps = torch.tensor([optimizer._steps], dtype=torch.int64, device=optimizer.flat_param.device)
    dist.broadcast(steps, src=src)
    optimizer._steps = int(steps.item())
    lo = optimizer.flat_param.data_ptr()
    hi = lo + optimizer.flat_param.numel() * 4
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            if lo <= t.data_ptr() < hi or t.numel() == 0:
                continue
            if t.is_contiguous():
                dist.broadcast(t.data, src=src)
            else:                                   # channels_last frozen conv weights: broadcast the bytes as they lie
                buf = t.data.contiguous()
                dist.broadcast(buf, src=src)
                t.data.copy_(buf)
    # the broadcast wrote the flat buffer, not the parameter tensors: their `_version` did not move, so every cached
    # image of the old weights (pre-split forward images, data-gradient images) must be dropped by hand
    optimizer.invalidate_images()


def backward_losses(losses):
    """d(sum of the loss terms)/d(parameters) -- the reference's `losses.backward()` on the summed dict
    (tools/rcnn/train_net.py:60-63) -- without forming the sum: every term starts its own backward pass with gradient
    1, which is the same gradient, minus the 7 add kernels, their autograd nodes and ~0.1 ms of host time that sat
    between the last forward and the first backward kernel, where the device has nothing else queued."""
    from pet.lib.ops import _hip as H
    H.deferred.clear()                      # (leftovers of a backward pass that raised half-way)
    terms = [v for v in losses.values() if v.requires_grad]
    # the seeds: ONE cached scalar 1 per device instead of a fresh ones_like per term (eight fill kernels and
    # allocations queued exactly where the device waits for the host: the start of the backward pass)
    seeds = [H.unit_seed(v.device) if v.dim() == 0 and v.dtype == torch.float32 and v.is_cuda else torch.ones_like(v)
             for v in terms]
    torch.autograd.backward(terms, seeds)


def reduce_losses(losses):
    """One collective for all logged scalars (the reference issues one blocking all_reduce per key)."""
    keys = sorted(losses.keys())
    vec = torch.stack([losses[k].detach().float().reshape(()) for k in keys])
    if world() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        vec = vec / world()
    return dict(zip(keys, vec.tolist()))
