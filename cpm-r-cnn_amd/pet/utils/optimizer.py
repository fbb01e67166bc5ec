"""SGD for the CPM R-CNN schedule (counterpart of pet/utils/optimizer.py:7-78) on a flat parameter buffer.

Same three parameter groups as the reference (weights: weight decay; biases: lr x2 when BIAS_DOUBLE_LR and no
weight decay unless BIAS_WEIGHT_DECAY; GroupNorm affine: WEIGHT_DECAY_GN), each carrying `lr_scale` so that
LearningRateScheduler drives them the same way.  What differs is the mechanics: every trainable tensor is
re-pointed into ONE fp32 buffer (parameters), with matching flat gradient and momentum buffers, laid out in
reverse forward order so that gradient chunks complete front-to-back during backward.  The update of all 196
tensors is then one launch of cpm_sgd_step, and the data-parallel all-reduce works on a few large contiguous
chunks (pet.utils.parallel) instead of per-tensor buckets."""
import ctypes
import os

import torch
import torch.nn as nn

from pet.lib.ops import _hip as H


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, named_params, groups_cfg, momentum):
        """named_params: list of (name, param, group_index) in flat-buffer order."""
        params = [p for _, p, _ in named_params]
        device = params[0].device
        align = 64                                 # every tensor starts on a 256-byte boundary (16-B vector loads)
        total = sum((p.numel() + align - 1) // align * align for p in params)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_mom = torch.zeros(total, dtype=torch.float32, device=device)
        begins, ends, gidx = [], [], []
        off = 0
        for _, p, gi in named_params:
            n = p.numel()
            self._repoint(p, off, n)
            begins.append(off)
            ends.append(off + n)
            gidx.append(gi)
            off += (n + align - 1) // align * align
        self.names = [n for n, _, _ in named_params]
        self._flat_order = params
        self._seg_index = {id(p): si for si, p in enumerate(params)}
        self.seg_begin = torch.tensor(begins, dtype=torch.int64, device=device)
        self.seg_end = torch.tensor(ends, dtype=torch.int64, device=device)
        self.seg_group = torch.tensor(gidx, dtype=torch.int64, device=device)
        self.seg_lr = torch.zeros(len(begins), dtype=torch.float32, device=device)
        self.seg_wd = torch.zeros(len(begins), dtype=torch.float32, device=device)
        # segment of every 64-element block (-1 in the alignment gaps): the SGD kernel looks its segment up instead
        # of searching
        table = torch.full((total // align,), -1, dtype=torch.int32)
        for si, (b, e) in enumerate(zip(begins, ends)):
            table[b // align:(e + align - 1) // align] = si
        self.block_seg = table.to(device)
        self.total, self.momentum, self._steps, self._last = total, float(momentum), 0, None
        self._stepped_end, self._w4_fresh = 0, False
        self.clear_grads_in_step, self._grads_cleared, self._write_gen_at_step = False, False, -1
        self.grad_scale = 1.0
        groups = []
        for gi, g in enumerate(groups_cfg):
            groups.append(dict(params=[p for _, p, k in named_params if k == gi], lr=0.0,
                               weight_decay=g["weight_decay"], lr_scale=g["lr_scale"], momentum=momentum))
        super().__init__([g for g in groups if len(g["params"])] or groups, dict(lr=0.0, momentum=momentum,
                                                                                weight_decay=0.0, lr_scale=1))
        self._group_of = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                self._group_of[id(p)] = gi
        self._seg_pg = [self._group_of[id(p)] for p in params]
        self.seg_pg = torch.tensor(self._seg_pg, dtype=torch.int32, device=device)

    def _view(self, flat, p, off, n):
        v = flat[off:off + n]
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
            k, c, r, s = p.shape
            return v.view(k, r, s, c).permute(0, 3, 1, 2)            # KRSC bytes, logical [K,C,R,S]
        return v.view(p.shape)

    def _repoint(self, p, off, n):
        dst = self._view(self.flat_param, p, off, n)
        dst.copy_(p.data)
        p.data = dst
        p.grad = self._view(self.flat_grad, p, off, n)
        if p.dim() in (1, 2, 4):
            # conv / Linear weights, biases and GroupNorm affine parameters: the HIP backward kernels accumulate straight into
            # this slice of the flat gradient buffer (no temporary, no autograd add)
            p._cpm_grad_sink = p.grad
            p._cpm_uses = 0

    # ---- serialisation in torch.optim.SGD's layout -----------------------------------------------------------
    # The reference saves torch.optim.SGD.state_dict() (pet/utils/checkpointer.py:128-133): parameters numbered group by
    # group (weights, biases, GroupNorm affine) in model.named_parameters() order, each with a 'momentum_buffer'.  The
    # flat buffers here are laid out in REVERSE registration order, so the canonical numbering walks each group
    # backwards.  Checkpoints therefore move between the two implementations unchanged.
    def _canonical(self):
        out = []
        for g in self.param_groups:
            out.append([self._seg_index[id(p)] for p in reversed(g["params"])])
        return out

    def state_dict(self):
        self.wait()
        begins, ends = self.seg_begin.tolist(), self.seg_end.tolist()
        state, groups, idx = {}, [], 0
        for g, segs in zip(self.param_groups, self._canonical()):
            ids = []
            for si in segs:
                if self._steps > 0:
                    p = self._flat_order[si]
                    buf = self._view(self.flat_mom, p, begins[si], ends[si] - begins[si])
                    buf = buf.detach().cpu().contiguous()
                    state[idx] = {"momentum_buffer": buf.reshape(getattr(p, "_cpm_abi_shape", buf.shape))}
                ids.append(idx)
                idx += 1
            meta = {k: v for k, v in g.items() if k != "params"}
            meta["params"] = ids
            groups.append(meta)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        self.wait()
        saved = [g for g in state_dict["param_groups"] if len(g["params"])]
        mine = self._canonical()
        if [len(g["params"]) for g in saved] != [len(s) for s in mine]:
            raise ValueError("optimizer state does not fit: saved groups %s, model groups %s"
                             % ([len(g["params"]) for g in saved], [len(s) for s in mine]))
        begins, ends = self.seg_begin.tolist(), self.seg_end.tolist()
        state = state_dict["state"]
        loaded = 0
        self.flat_mom.zero_()
        for g, sg, segs in zip(self.param_groups, saved, mine):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
            for idx, si in zip(sg["params"], segs):
                st = state.get(idx, state.get(str(idx)))
                if not st or st.get("momentum_buffer") is None:
                    continue
                p = self._flat_order[si]
                mb = st["momentum_buffer"]
                if tuple(mb.shape) == tuple(getattr(p, "_cpm_abi_shape", ())):
                    mb = mb.reshape(p.shape)                # stored [K, C*H*W] (reference layout) -> [K,C,H,W]
                if tuple(mb.shape) != tuple(p.shape):
                    raise ValueError("momentum buffer %d has shape %s, parameter %s has %s"
                                     % (idx, tuple(mb.shape), self.names[si], tuple(p.shape)))
                self._view(self.flat_mom, p, begins[si], ends[si] - begins[si]).copy_(mb)
                loaded += 1
        # momentum buffers that exist mean "not the first step" (torch.optim.SGD initialises buf = d on first use;
        # with a zero buffer momentum*0 + d is the same value, so a partially filled state is still exact)
        self._steps = 1 if loaded else 0

    def wait(self):
        """the current stream waits for an update queued beside it (overlap_next_forward): call before reading parameters
        or optimizer state with anything but this package's ops (they wait by themselves)"""
        H.wait_pending_sgd(self.flat_param.device if self.flat_param.is_cuda else None)

    def zero_grad(self, set_to_none=False):
        # one memset; .grad views stay attached.  With `clear_grads_in_step` (a trainer's choice: gradients are then zero
        # after step(), unlike torch.optim's) the SGD kernel has already cleared every element behind its use.
        # A backward pass that ran since then outside a trainer step (the warm-up iterations of a hipGraph capture, a
        # gradient probe) has written gradients again: every in-place route counts a use in its forward
        # (conv._note_use), and the memset is skipped only when nothing was counted since step().
        from pet.lib.ops import conv as C
        if not (self.clear_grads_in_step and self._grads_cleared
                and C.grad_write_generation() == self._write_gen_at_step):
            self.wait()                              # (the update reads the gradients it is about to clear)
            self.flat_grad.zero_()
        self._grads_cleared = False
        for g in self.param_groups:
            for p in g["params"]:
                if hasattr(p, "_cpm_uses"):
                    p._cpm_uses = 0

    @torch.no_grad()
    def step_range(self, begin, end):
        """SGD update of flat elements [begin, end) (64-element boundaries) on the CURRENT stream: a chunk whose
        gradients are complete (pet.utils.parallel.FlatGradReducer launches it on its side stream, behind the chunk's
        all-reduce); `step()` then covers what is left.  Ranges of one step must be issued front to back."""
        from pet.lib.ops import conv as C
        assert begin == self._stepped_end and begin % 64 == 0 and end % 64 == 0 and end <= self.total
        if end == begin:
            return
        ng = len(self.param_groups)
        lr = (ctypes.c_float * ng)(*[float(g["lr"]) for g in self.param_groups])
        wd = (ctypes.c_float * ng)(*[float(g["weight_decay"]) for g in self.param_groups])
        w4 = C._W4 and C.bf16x3()
        if w4 and getattr(self, "flat_w4", None) is None:
            self.flat_w4 = torch.empty_like(self.flat_param)
        self._w4_fresh = w4 if begin == 0 else (self._w4_fresh and w4)
        with torch.cuda.device(self.flat_param.device):
            # under bf16x3 the pre-split image of every parameter (the forward convs' weight operand, conv.w4_of) leaves
            # the same pass: one more write stream of the SGD kernel instead of a pass per weight
            rc = H.lib().cpm_sgd_step_range(H.ptr(self.flat_param), H.ptr(self.flat_grad), H.ptr(self.flat_mom),
                                            H.ptr(self.block_seg), H.ptr(self.seg_end), H.ptr(self.seg_pg), lr, wd, ng,
                                            H.c_int64(begin), H.c_int64(end), H.f(self.momentum),
                                            H.f(self.grad_scale), int(self._steps == 0),
                                            H.ptr(self.flat_w4) if w4 else None, int(self.clear_grads_in_step),
                                            H.stream())
        H.check(rc, "sgd_step")
        self._stepped_end = end

    @torch.no_grad()
    def step(self, closure=None):
        # overlap_next_forward (a training loop's choice, off by default): the update -- HBM streaming, nothing for the
        # matrix cores -- and the weight-image transform behind it go to the optimizer stream; the next forward pass
        # starts at once and waits where it first touches a trainable tensor (H.wait_pending_sgd), i.e. behind the frozen
        # stem / layer1.  Readers of parameters outside this package's ops call wait() first.
        ov = bool(getattr(self, "overlap_next_forward", False)) and self.flat_param.is_cuda
        self._opt_stream = None
        if ov:
            dev = self.flat_param.device
            H.wait_pending_sgd(dev)                  # (an earlier update nobody waited for: keep them in order)
            st = H.optimizer_stream(dev)
            H.fork(H._raw_stream(dev.index if dev.index is not None else torch.cuda.current_device()), st.cuda_stream)
            self._opt_stream = st
            with H.use_stream(st.cuda_stream):
                if self._stepped_end < self.total:
                    self.step_range(self._stepped_end, self.total)
            ev = torch.cuda.Event()
            ev.record(st)
            H.set_pending_sgd_event(ev, dev)
        elif self._stepped_end < self.total:
            self.step_range(self._stepped_end, self.total)
        self._stepped_end = 0
        self._steps += 1
        self._grads_cleared = bool(self.clear_grads_in_step)
        from pet.lib.ops import conv as C
        self._write_gen_at_step = C.grad_write_generation()
        self._mark_w4(self._w4_fresh)
        self._refresh_dgrad_weights()

    def _mark_w4(self, fresh):
        """Parameters whose pre-split image the SGD kernel has just written: conv / Linear weights (numel % 4 == 0) get
        the view of the image buffer at their own offset, stamped with their `_version` (conv.w4_of reads it; an
        in-place modification by anything else moves `_version` and w4_of re-splits into the same view)."""
        if not fresh:
            # the SGD kernel writes the parameters through raw pointers (no `_version` bump): EVERY cached image is
            # stale after a step that did not write it -- the ones conv.w4_of split by itself before the first
            # step or under another arithmetic included, not only the views this method handed out
            for p in self._flat_order:
                if hasattr(p, "_cpm_w4"):
                    p._cpm_w4_version = -1
            self._w4_marked = False
            return
        if not getattr(self, "_w4_marked", False):
            begins = self.seg_begin.tolist()
            for si, p in enumerate(self._flat_order):
                if p.dim() in (2, 4) and p.numel() % 4 == 0:
                    p._cpm_w4 = self.flat_w4[begins[si]:begins[si] + p.numel()]
                    p._cpm_w4_ptr = p.data_ptr()
            self._w4_marked = True
        for p in self._flat_order:
            if hasattr(p, "_cpm_w4"):
                p._cpm_w4_version = p._version

    def invalidate_images(self):
        """Call after ANY write to the flat parameter buffer that does not go through the parameters' own tensors
        (a broadcast or copy into `flat_param`, a raw-pointer kernel): views of one storage keep separate version
        counters, so neither the pre-split forward images (conv.w4_of, keyed on `_version`) nor the data-gradient
        images (conv._prepared_wt) notice such a write by themselves.  The images are rebuilt on their next use."""
        for p in self._flat_order:
            if hasattr(p, "_cpm_w4"):
                p._cpm_w4_version = -1
            if hasattr(p, "_cpm_wt_version"):
                p._cpm_wt_version = -1
        self._w4_marked = False

    def _refresh_dgrad_weights(self):
        """Every conv weight that has been used by a data-gradient call (pet.lib.ops.conv._prepared_wt registers it
        with its geometry) gets its data-gradient image rebuilt here, right behind the SGD kernel, for ALL weights in
        ONE launch -- instead of one small transform launch in front of each of the ~100 data-gradient calls of the
        next backward pass."""
        params = [p for p in self._flat_order if getattr(p, "_cpm_wt_desc", None) is not None]
        if not params:
            return
        if getattr(self, "_wt_params", None) is None or len(self._wt_params) != len(params):
            if getattr(self, "flat_wt", None) is None:
                self.flat_wt = torch.empty_like(self.flat_param)
            begins = self.seg_begin.tolist()
            rows, tiles = [], 0
            for p in params:
                groups, kg, rs, cg, scale_ptr = p._cpm_wt_desc
                off = begins[self._seg_index[id(p)]]
                # scale_ptr: the frozen per-channel factor behind the conv (AffineChannel2d.weight: never reallocated),
                # folded into the image (cpm_wt_desc.k_scale)
                rows.append((off, off, groups, kg, rs, cg, tiles, scale_ptr))
                tiles += ((cg + 31) // 32) * ((kg + 31) // 32) * rs * groups
            import numpy as np
            tab = np.zeros(len(rows), dtype=[("src", "<i8"), ("dst", "<i8"), ("g", "<i4"), ("kg", "<i4"), ("rs", "<i4"),
                                             ("cg", "<i4"), ("t0", "<i8"), ("scale", "<i8")])
            for i, r in enumerate(rows):
                tab[i] = r
            self._wt_table = torch.from_numpy(tab.view(np.uint8).copy()).to(self.flat_param.device)
            self._wt_tiles = tiles
            self._wt_params = params
            for p in params:
                off = begins[self._seg_index[id(p)]]
                p._cpm_wt = self.flat_wt[off:off + p.numel()]
        # The images are first read by the NEXT backward pass: the transform (0.36 ms for R-50) runs on the second
        # stream, beside the next step's forward pass, and the compute stream waits for it at its first data-gradient
        # call (pet.lib.ops.conv._prepared_wt).  The side stream starts behind the SGD kernel.
        from pet.lib.ops import conv as C
        dev = self.flat_param.device
        side = C.wgrad_stream(dev) if os.environ.get("CPM_WT_ON_SIDE", "1") != "0" else None
        own = getattr(self, "_opt_stream", None)         # step() put the update on the optimizer stream: stay behind it
        if own is not None:
            side = own
        # under bf16x3 the images are written pre-split (weights with K / groups % 4 == 0; conv._prepared_call)
        w4 = C._W4 and C.bf16x3()
        transform = H.lib().cpm_weights_to_dgrad_batched_w4 if w4 else H.lib().cpm_weights_to_dgrad_batched
        with H.guard(dev):
            if side is not None:
                if own is None:
                    main_raw = H._raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
                    H.fork(main_raw, side.cuda_stream)
                with H.use_stream(side.cuda_stream):
                    rc = transform(H.ptr(self._wt_table), len(self._wt_params), H.c_int64(self._wt_tiles),
                                   H.ptr(self.flat_param), H.ptr(self.flat_wt), H.stream())
                ev = torch.cuda.Event()
                ev.record(side)
                C.set_pending_wt_event(ev, dev)
            else:
                rc = transform(H.ptr(self._wt_table), len(self._wt_params), H.c_int64(self._wt_tiles),
                               H.ptr(self.flat_param), H.ptr(self.flat_wt), H.stream())
        H.check(rc, "weights_to_dgrad_batched")
        for p in self._wt_params:
            p._cpm_wt_fmt = 1 if (w4 and p._cpm_wt_desc[1] % 4 == 0) else 0
            p._cpm_wt_version = p._version
            sc = getattr(p, "_cpm_wt_scale", None)
            if sc is not None:
                p._cpm_wt_scale_version = sc._version


class Optimizer(object):
    def __init__(self, model, solver, local_rank=0, frozen_lr_keys=()):
        """frozen_lr_keys: parameters whose name contains one of these strings form a fourth group with lr_scale 0 and no
        weight decay -- their gradients are computed, their values stay (bench.py holds config #5's offset predictors
        at their initialisation this way; the reference has no such group: pet/utils/optimizer.py:40-65)."""
        self.model, self.solver, self.local_rank = model, solver, local_rank
        self.frozen_lr_keys = tuple(frozen_lr_keys)

    def build(self):
        S = self.solver
        if S.OPTIMIZER != "SGD":
            raise ValueError("only SOLVER.OPTIMIZER == 'SGD' is built (the CPM R-CNN schedule)")
        gn_names = set()
        for name, m in self.model.named_modules():
            if isinstance(m, nn.GroupNorm):
                gn_names.update((name + ".weight", name + ".bias"))
        named = []
        for key, p in self.model.named_parameters():
            if not p.requires_grad:
                continue
            gi = 1 if "bias" in key else (2 if key in gn_names else 0)      # optimizer.py:30-38 order of tests
            if any(f in key for f in self.frozen_lr_keys):
                gi = 3
            named.append((key, p, gi))
        named.reverse()                      # reverse registration (~forward) order: backward fills front to back
        groups = [dict(weight_decay=S.WEIGHT_DECAY, lr_scale=1),
                  dict(weight_decay=S.WEIGHT_DECAY if S.BIAS_WEIGHT_DECAY else 0, lr_scale=S.BIAS_DOUBLE_LR + 1),
                  dict(weight_decay=S.WEIGHT_DECAY_GN * S.WEIGHT_DECAY, lr_scale=1)]
        if self.frozen_lr_keys:
            groups.append(dict(weight_decay=0.0, lr_scale=0))
        return FlatSGD(named, groups, S.MOMENTUM)
