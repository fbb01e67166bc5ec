"""GPU: the device-resident RoI lists of the training step (csrc/roi_lists.hip, pet/lib/ops/roi_lists.py) against the
host-index formulations they replace -- which the other GPU tests hold to the reference's per-image BoxList code and to
the reference-generated fixtures.  Integer / index / copy work: every comparison is exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand_boxes(rng, n, w, h, lo, hi):
    cx, cy = rng.uniform(0, w, n), rng.uniform(0, h, n)
    bw, bh = rng.uniform(lo, hi, n), rng.uniform(lo, hi, n)
    b = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
    return np.clip(b, 0, [w - 1, h - 1, w - 1, h - 1]).astype(np.float32)


def _proposal_like(rng, n_img, per_img, n_gt, w=640, h=480):
    """per image: gts, jittered copies of them (positives / between), random boxes (negatives)"""
    from pet.utils.data.structures.bounding_box import BoxList
    props, targets = [], []
    for i in range(n_img):
        g = _rand_boxes(rng, n_gt[i], w, h, 40, 200)
        jit = g[rng.integers(0, n_gt[i], per_img[i] // 3)] + rng.normal(0, 6, (per_img[i] // 3, 4)).astype(np.float32)
        neg = _rand_boxes(rng, per_img[i] - len(jit) - n_gt[i], w, h, 10, 300)
        b = np.concatenate([jit, neg, g]).astype(np.float32)
        b = np.clip(b, 0, [w - 1, h - 1, w - 1, h - 1]).astype(np.float32)
        p = BoxList(torch.from_numpy(b).cuda(), (w, h))
        p.add_field("objectness", torch.from_numpy(rng.uniform(0, 1, len(b)).astype(np.float32)).cuda())
        t = BoxList(torch.from_numpy(g).cuda(), (w, h))
        t.add_field("labels", torch.from_numpy(rng.integers(1, 81, n_gt[i])).cuda())
        props.append(p)
        targets.append(t)
    return props, targets


@pytest.mark.parametrize("top_k,quant", [(2000, 0), (300, 0), (300, 64), (5, 0)])
def test_proposals_finalize_equals_host_index_path(top_k, quant):
    """cpm_proposals_finalize against RPNPostProcessor.finish_fused (host index lists + torch.topk + cat) on the same
    NMS output: the same proposals in the same order, gts appended per image.  `quant` coarsens the scores so that
    the top-k boundary falls inside a run of equal scores (ties go to the lowest index, as torch.topk does)."""
    from pet.lib.ops import roi_lists as RL
    from pet.lib.ops import nms_segments
    from pet.rcnn.modeling.rpn.inference import RPNPostProcessor
    rng = np.random.default_rng(top_k + quant)
    N, L, k = 2, 4, 400
    sel = RPNPostProcessor(2000, 250, 0.7, 0, fpn_post_nms_top_n=top_k, fpn_post_nms_per_batch=True).train()
    offsets, owner, boxes, scores = [0], [], [], []
    for lvl in range(L):
        for n in range(N):
            m = k if lvl < 3 else 37
            boxes.append(_rand_boxes(rng, m, 640, 480, 16, 200))
            s = rng.uniform(0, 1, m).astype(np.float32)
            scores.append(np.round(s * quant) / quant if quant else s)
            offsets.append(offsets[-1] + m)
            owner.append(n)
    all_boxes = torch.from_numpy(np.concatenate(boxes)).cuda()
    all_scores = torch.from_numpy(np.concatenate(scores).astype(np.float32)).cuda()
    keep, counts = nms_segments(all_boxes, all_scores, None, offsets, 0.7, 0)
    _, targets = _proposal_like(rng, N, [10, 10], [3, 5])
    sizes = [(640, 480)] * N
    st = dict(num_levels=L, N=N, dev=all_boxes.device, sizes=sizes, offsets=offsets, owner=owner, all_boxes=all_boxes,
              all_scores=all_scores, keep=keep, counts=counts)
    st["counts_h"] = counts.cpu()
    st["event"] = torch.cuda.Event()
    st["event"].record()
    want = sel.finish_fused(st, targets)
    got = sel.finish_device(st, targets)
    c = got.counts.cpu().tolist()
    assert c[:-1] == [len(b) for b in want] and c[-1] == sum(len(b) for b in want)
    tot = c[-1]
    assert tot <= got.capacity
    wb = torch.cat([b.bbox for b in want])
    wo = torch.cat([b.get_field("objectness") for b in want])
    assert torch.equal(got.boxes[:tot], wb) and torch.equal(got.obj[:tot], wo)
    img = np.repeat(np.arange(N), c[:-1])
    assert np.array_equal(got.img[:tot].cpu().numpy(), img)
    assert bool((got.img[tot:] == -1).all())


@pytest.mark.parametrize("batch,frac,max_grid", [(512, 0.25, 96), (512, 0.25, 20), (64, 0.5, 8), (4096, 1.0, 4096)])
def test_roi_sample_equals_match_label_sampler(batch, frac, max_grid):
    """cpm_roi_sample against cpm_match_rois + the label arithmetic + cpm_sample_pos_neg (same seed) + boolean-mask
    compaction: the same sample rows in the same order, the same labels; and its positives list against
    keep_only_positive_boxes' definition (labels > 0, a subset of max_grid per image when there are more) with the
    first grid stage's matched gt box."""
    import pet.lib.ops as ops
    from pet.lib.ops import roi_lists as RL
    rng = np.random.default_rng(batch + max_grid)
    n_img = 3
    props, targets = _proposal_like(rng, n_img, [1500, 900, 40], [7, 3, 1])
    inp = RL.from_boxlists(props)
    gt_all, gt_labels, gt_off, off_h = RL.gt_pack(targets)
    hi, lo, seed, seed_g, grid_hi = 0.5, 0.5, 123456789012345, 987654321, 0.6
    s, p, counts = RL.roi_sample(inp, gt_all, gt_labels, gt_off, hi, lo, batch, frac, seed, max_grid, seed_g, grid_hi)
    c = counts.cpu().tolist()
    assert c[-1] == 0
    sc, pc = c[:n_img + 1], c[n_img + 1:2 * (n_img + 1)]
    # the formulation it replaces
    cnt = [len(b) for b in props]
    img = torch.from_numpy(np.repeat(np.arange(n_img), cnt).astype(np.int32)).cuda()
    base = torch.from_numpy(np.asarray(off_h)[np.repeat(np.arange(n_img), cnt)]).cuda()
    matched, best = ops.match_rois(inp.boxes, img, gt_all, gt_off, hi, lo, False)
    lab = gt_labels[matched.clamp(min=0) + base]
    lab = torch.where(matched == -1, 0, lab)
    lab = torch.where(matched == -2, -1, lab)
    pos, neg, quota = ops.sample_pos_neg(lab, cnt, batch, frac, seed=seed)
    take = (pos | neg)
    idx = torch.nonzero(take).squeeze(1)
    want_counts = np.bincount(img[idx].cpu().numpy(), minlength=n_img).tolist()
    assert sc[:-1] == want_counts and sc[-1] == sum(want_counts)
    S = sc[-1]
    assert torch.equal(s.boxes[:S], inp.boxes[idx]) and torch.equal(s.obj[:S], inp.obj[idx])
    assert torch.equal(s.labels[:S], lab[idx]) and torch.equal(s.img[:S], img[idx])
    assert torch.equal(s.rois5[:S, 1:], inp.boxes[idx]) and torch.equal(s.rois5[:S, 0], img[idx].float())
    assert bool((s.labels[S:] == -100).all()) and bool((s.img[S:] == -1).all())
    # positives
    P = pc[-1]
    sl = s.labels[:S].cpu().numpy()
    simg = s.img[:S].cpu().numpy()
    src = p.src[:P].cpu().numpy()
    assert np.all(np.diff(src) > 0) and np.all(sl[src] > 0)
    for i in range(n_img):
        n_pos_i = int(((sl > 0) & (simg == i)).sum())
        assert pc[i] == min(n_pos_i, max_grid), (i, pc[i], n_pos_i)
        if n_pos_i <= max_grid:
            assert np.array_equal(src[simg[src] == i], np.flatnonzero((sl > 0) & (simg == i)))
    assert torch.equal(p.boxes[:P], s.boxes[:S][p.src[:P]]) and torch.equal(p.img[:P], s.img[:S][p.src[:P]])
    m0, iou0 = ops.match_rois(p.boxes[:P], p.img[:P], gt_all, gt_off, grid_hi, grid_hi, False)
    gbase = torch.from_numpy(np.asarray(off_h)).cuda()[p.img[:P].long()]
    assert torch.equal(p.gt[:P], gt_all[m0.clamp(min=0) + gbase]) and torch.equal(p.iou[:P], iou0)
    # a different grid seed draws a different subset of the same size
    if any(int(((sl > 0) & (simg == i)).sum()) > max_grid for i in range(n_img)):
        _, p2, c2 = RL.roi_sample(inp, gt_all, gt_labels, gt_off, hi, lo, batch, frac, seed, max_grid, seed_g + 1, grid_hi)
        assert c2.cpu().tolist() == c
        assert not torch.equal(p2.src[:P], p.src[:P])


def test_roi_sample_reports_oversized_images():
    from pet.lib.ops import roi_lists as RL
    rng = np.random.default_rng(5)
    props, targets = _proposal_like(rng, 1, [RL.roi_sample_max_rows() + 1], [4])
    inp = RL.from_boxlists(props)
    gt_all, gt_labels, gt_off, _ = RL.gt_pack(targets)
    # the host refuses a list whose capacity could overflow an image's LDS share BEFORE anything is queued (the
    # kernel's own status word would only be read after the cls head had run on an unwritten sample)
    with pytest.raises(RuntimeError, match="CPM_DEVICE_LISTS=0"):
        RL.roi_sample(inp, gt_all, gt_labels, gt_off, 0.5, 0.5, 512, 0.25, 1)
    # the kernel still reports it for callers of the C ABI that skip the check
    big = inp.capacity
    inp.capacity = RL.roi_sample_max_rows()
    try:
        _, _, counts = RL.roi_sample(inp, gt_all, gt_labels, gt_off, 0.5, 0.5, 512, 0.25, 1)
    finally:
        inp.capacity = big
    assert counts.cpu().tolist()[-1] == 1


def test_stage_advance_equals_host_index_path():
    """cpm_stage_advance against the index list _forward_train_cascade_fused builds on the host."""
    from pet.lib.ops import roi_lists as RL
    rng = np.random.default_rng(11)
    n_img, cnt, n_gt = 3, [70, 0, 45], [4, 2, 6]
    _, targets = _proposal_like(rng, n_img, [20, 20, 20], n_gt)
    gt_all, _, gt_off, off_h = RL.gt_pack(targets)
    R, G = sum(cnt), off_h[-1]
    refined = torch.from_numpy(_rand_boxes(rng, R, 640, 480, 10, 200)).cuda()
    keep = torch.from_numpy(rng.uniform(0, 1, R) < 0.8).cuda()
    img_h = np.repeat(np.arange(n_img), cnt)
    m2_h = np.array([rng.integers(-2, n_gt[i]) for i in img_h])
    m2 = torch.from_numpy(m2_h).cuda()
    iou2 = torch.from_numpy(rng.uniform(0, 1, R).astype(np.float32)).cuda()
    img = torch.from_numpy(img_h.astype(np.int32)).cuda()
    src = torch.from_numpy(rng.permutation(1000)[:R]).cuda()
    for src_in in (src, None):
        out = RL.stage_advance(refined, keep, m2, iou2, img, src_in, n_img, 1000, gt_all, gt_off, G, None)
        keep_h = (keep.cpu().numpy() & (m2_h >= 0))
        off = np.concatenate([[0], np.cumsum(cnt)])
        index, counts = [], []
        for i in range(n_img):
            kept = np.flatnonzero(keep_h[off[i]:off[i + 1]]) + off[i]
            index.extend(kept.tolist())
            index.extend(range(R + off_h[i], R + off_h[i + 1]))
            counts.append(len(kept) + n_gt[i])
        index = torch.tensor(index).cuda()
        base = torch.from_numpy(np.asarray(off_h)[img_h]).cuda()
        c = out.counts.cpu().tolist()
        assert c == counts + [sum(counts)]
        T = c[-1]
        src_rows = src if src_in is not None else torch.arange(R).cuda()
        assert torch.equal(out.boxes[:T], torch.cat([refined, gt_all])[index])
        assert torch.equal(out.gt[:T], torch.cat([gt_all[m2.clamp(min=0) + base], gt_all])[index])
        assert torch.equal(out.iou[:T], torch.cat([iou2, torch.ones(G).cuda()])[index])
        assert torch.equal(out.src[:T], torch.cat([src_rows, torch.arange(1000, 1000 + G).cuda()])[index])
        assert np.array_equal(out.img[:T].cpu().numpy(), np.repeat(np.arange(n_img), counts))
        assert torch.equal(out.rois5[:T, 1:], out.boxes[:T]) and torch.equal(out.rois5[:T, 0], out.img[:T].float())
        assert bool((out.img[T:] == -1).all())


def test_rescore_gather_equals_get_full_sample_boxes():
    """cpm_rescore_gather against get_full_sample_boxes on per-image BoxLists (cls negatives, then refined RoIs)."""
    from pet.lib.ops import roi_lists as RL
    from pet.rcnn.modeling.grid_cascade_rcnn.grid_cascade_rcnn import get_full_sample_boxes
    from pet.utils.data.structures.bounding_box import BoxList
    rng = np.random.default_rng(17)
    n_img, sc, gc = 3, [300, 5, 128], [20, 3, 0]
    S, Gr = sum(sc), sum(gc)
    sample = RL.RoIList(S + 7, n_img, [(640, 480)] * n_img,
                        boxes=torch.from_numpy(_rand_boxes(rng, S + 7, 640, 480, 10, 200)).cuda(),
                        obj=torch.from_numpy(rng.uniform(0, 1, S + 7).astype(np.float32)).cuda(),
                        labels=torch.from_numpy(rng.integers(-1, 4, S + 7)).cuda(),
                        counts=torch.tensor(sc + [S], dtype=torch.int32).cuda())
    n_first = 40
    first_src = torch.from_numpy(rng.permutation(S)[:n_first]).cuda()
    last = RL.RoIList(Gr + 3, n_img, None, boxes=torch.from_numpy(_rand_boxes(rng, Gr + 3, 640, 480, 10, 200)).cuda(),
                      src=torch.from_numpy(rng.integers(0, n_first + 9, Gr + 3)).cuda(),
                      counts=torch.tensor(gc + [Gr], dtype=torch.int32).cuda())
    out = RL.rescore_gather(sample, last, first_src, n_first, S + Gr)
    # BoxList formulation
    obj_all = torch.cat([sample.obj[first_src], torch.ones(9).cuda()])
    cls_lists, grid_lists, o, g = [], [], 0, 0
    for i in range(n_img):
        c = BoxList(sample.boxes[o:o + sc[i]], (640, 480))
        c.add_field("objectness", sample.obj[o:o + sc[i]])
        c.add_field("labels", sample.labels[o:o + sc[i]])
        r = BoxList(last.boxes[g:g + gc[i]], (640, 480))
        r.add_field("objectness", obj_all[last.src[g:g + gc[i]]])
        r.add_field("labels", torch.ones(gc[i], dtype=torch.int64).cuda())
        cls_lists.append(c)
        grid_lists.append(r)
        o += sc[i]
        g += gc[i]
    want = get_full_sample_boxes(cls_lists, grid_lists)
    c = out.counts.cpu().tolist()
    assert c == [len(b) for b in want] + [sum(len(b) for b in want)]
    T = c[-1]
    assert torch.equal(out.boxes[:T], torch.cat([b.bbox for b in want]))
    assert torch.equal(out.obj[:T], torch.cat([b.get_field("objectness") for b in want]))


def test_device_list_head_equals_boxlist_head():
    """The whole training forward of the CPM head on packed device lists (_forward_train_lists) against the BoxList
    formulation (fused glue with host index lists), with sample budgets so large that no sampler draws: the same
    RoI sets at every stage, the same 6 losses, the same feature gradients (to the run-to-run noise of the float
    atomics in the heads' split-K / GroupNorm / RoIAlign reductions)."""
    from test_gpu_model import CPM_OPTS, synthetic_batch
    from detfill import det_fill_
    from pet.lib.ops import roi_lists as RL
    from pet.rcnn.core import config
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.data.structures.image_list import to_image_list
    from pet.utils.net import convert_bn2affine_model
    config.reset_cfg()
    config.merge_cfg_from_list(CPM_OPTS + ["GRID_RCNN.BATCH_SIZE_PER_IMAGE", 2048, "GRID_RCNN.POSITIVE_FRACTION", 1.0,
                                           "GRID_RCNN.MAX_SAMPLE_NUM_GRID", 2048])
    try:
        m = convert_bn2affine_model(Generalized_RCNN(is_train=True))
        det_fill_(m)
        m = m.cuda().to(memory_format=torch.channels_last).train()
        assert m.RPN.roi_heads_take_lists
        images, targets = synthetic_batch(2, 256, 320, 6, seed=3)
        targets = [t.to("cuda") for t in targets]
        head = m.Grid_Cascade_RCNN
        with torch.no_grad():
            feats = m.Conv_Body_FPN(m.Conv_Body(images.cuda().contiguous(memory_format=torch.channels_last)))
        feats = [f.detach().requires_grad_(True) for f in feats]
        m.RPN.roi_heads_take_lists = False
        with torch.no_grad():
            props, _ = m.RPN(to_image_list(images.cuda()), feats, targets)
        m.RPN.roi_heads_take_lists = True
        # keep the per-image proposal count small enough that positives stay below every cap
        props = [p[torch.arange(min(len(p), 600), device="cuda")] for p in props]
        outs = []
        for mode in ("lists", "lists", "boxlists"):
            for f in feats:
                f.grad = None
            m.zero_grad(set_to_none=True)
            stage_rois = []
            hooks = [getattr(head, "Head_grid_%d" % s).register_forward_pre_hook(
                lambda mod, a: stage_rois.append(torch.cat([b.bbox for b in a[1]]).clone())) for s in range(3)]
            inp = RL.from_boxlists(props) if mode == "lists" else [p[torch.arange(len(p), device="cuda")] for p in props]
            _, result, losses = head(feats, inp, targets)
            for h in hooks:
                h.remove()
            sum(losses.values()).backward()
            outs.append((stage_rois, torch.cat([b.bbox for b in result]),
                         torch.cat([b.get_field("labels") for b in result]),
                         {k: float(v.detach()) for k, v in losses.items()}, dict(head.last_counts),
                         [f.grad.clone() for f in feats[:4]]))
        (sa, ra, la, lossa, ca, ga), (_, _, _, _, _, gn), (sb, rb, lb, lossb, cb, gb) = outs
        assert ca == cb, (ca, cb)
        for a, b in zip(sa, sb):
            assert torch.equal(a, b)
        assert torch.equal(ra, rb) and torch.equal(la, lb)
        assert set(lossa) == set(lossb) and len(lossa) == 6
        for k in lossa:
            assert abs(lossa[k] - lossb[k]) <= 1e-5 * abs(lossb[k]) + 1e-7, (k, lossa[k], lossb[k])
        scale = max(float(b.abs().max()) for b in gb)
        for lvl, (a, a2, b) in enumerate(zip(ga, gn, gb)):
            noise = float((a - a2).abs().max())
            assert float((a - b).abs().max()) <= 3 * noise + 2e-3 * scale + 1e-9, lvl
    finally:
        config.reset_cfg()


def test_sigmoid_multi_is_bitwise_torch_sigmoid():
    """The RPN scores must be the framework's sigmoid bit for bit: ties among saturated scores decide the top-k order."""
    import pet.lib.ops as ops
    g = torch.Generator().manual_seed(4)
    xs = [torch.randn(2, 3, 50, 84, generator=g).mul(6).cuda().contiguous(memory_format=torch.channels_last),
          torch.cat([torch.linspace(-110, 110, 70001), torch.tensor([0.0, -0.0, 88.7, -88.7, 1e-30, 17.3, -17.3])]).cuda(),
          torch.randn(5, generator=g).cuda()]
    outs = ops.sigmoid_multi(xs)
    for x, o in zip(xs, outs):
        want = torch.sigmoid(x)
        want = want.permute(0, 2, 3, 1).reshape(-1) if x.dim() == 4 else want.reshape(-1)
        assert torch.equal(o, want)


def test_rpn_decode_multi_and_labels_equal_single_level_ops():
    import pet.lib.ops as ops
    rng = np.random.default_rng(8)
    N, sizes = 2, [(640, 480), (600, 400)]
    As, ks = [5000, 1200, 300], [700, 700, 300]
    regs, idxs, ancs, offs = [], [], [], [0]
    for A, k in zip(As, ks):
        regs.append(torch.from_numpy(rng.normal(0, 0.5, (N, A, 4)).astype(np.float32)).cuda())
        idxs.append(torch.from_numpy(np.stack([rng.permutation(A)[:k] for _ in range(N)]).astype(np.int64)).cuda())
        ancs.append(torch.from_numpy(_rand_boxes(rng, A, 640, 480, 8, 300)).cuda())
        offs.append(offs[-1] + N * k)
    out = torch.empty((offs[-1], 4), dtype=torch.float32, device="cuda")
    ops.rpn_decode_multi(regs, idxs, ancs, offs[:-1], out, (1.0, 1.0, 1.0, 1.0), float(np.log(1000. / 16)), sizes)
    for l in range(3):
        want = ops.rpn_decode(regs[l], idxs[l], ancs[l], (1.0, 1.0, 1.0, 1.0), float(np.log(1000. / 16)), sizes)
        assert torch.equal(out[offs[l]:offs[l + 1]], want.view(-1, 4))
    matched = torch.from_numpy(rng.integers(-2, 5, 100000)).cuda()
    vis = torch.from_numpy(rng.uniform(0, 1, 100000) < 0.7).cuda()
    lab = (matched >= 0).float()
    lab = torch.where(matched == -2, -1.0, lab)
    assert torch.equal(ops.rpn_labels(matched, None, True), lab)
    assert torch.equal(ops.rpn_labels(matched, vis, True), torch.where(vis, lab, -1.0))
    assert torch.equal(ops.rpn_labels(matched, vis, False), torch.where(vis, (matched >= 0).float(), -1.0))


@pytest.mark.parametrize("R", [2, 3, 37, 700])
def test_l2_loss_fused_equals_reference_formulation(R):
    """cpm_l2_loss_pairs (value + gradient in one launch) against l2_loss -- the reference's nonzero() + x[pos_inds]
    formulation (pet/lib/ops/l2_loss.py:4-11) -- and its autograd gradient."""
    import pet.lib.ops as ops
    g = torch.Generator().manual_seed(R)
    iou = torch.rand(R, generator=g).cuda()
    iou[::5] = 1.0                                   # 1 - iou == 0: not a positive target
    iou[1::7] = 0.0
    x = torch.randn(R, 2, generator=g).cuda()
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    target = torch.stack([1 - iou, iou], dim=1)
    want = ops.l2_loss(xa, target)
    got = ops.l2_loss_fused(xb, iou=iou)
    got2 = ops.l2_loss_fused(x.clone(), target=target)
    assert abs(float(got.detach()) - float(want.detach())) <= 1e-5 * abs(float(want.detach())) + 1e-7
    assert abs(float(got2) - float(want.detach())) <= 1e-5 * abs(float(want.detach())) + 1e-7
    (want * 3).backward()
    (got * 3).backward()
    np.testing.assert_allclose(xb.grad.cpu().numpy(), xa.grad.cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("R,C", [(1, 81), (5, 2), (64, 81), (1024, 81), (3001, 7), (40, 300)])
def test_cross_entropy_fused_equals_the_framework_formulation(R, C):
    """cpm_softmax_ce (value + gradient of the cls / RSM heads' loss in one launch) against F.cross_entropy -- the
    reference's call (grid_cascade_rcnn/loss.py:103-112) -- in float64 and its autograd gradient; every third row carries
    the ignore index, as the rows beyond a capacity-sized sample's count do."""
    import torch.nn.functional as F
    import pet.lib.ops as ops
    g = torch.Generator().manual_seed(R * 1000 + C)
    x = (torch.randn(R, C, generator=g) * 4).cuda()
    lab = torch.randint(0, C, (R,), generator=g)
    if R > 2:
        lab[2::3] = -100
    lab = lab.cuda()
    xa = x.double().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    want = F.cross_entropy(xa, lab)
    got = ops.cross_entropy_fused(xb, lab)
    assert got.dtype == torch.float32 and got.dim() == 0
    assert abs(float(got.detach()) - float(want.detach())) <= 2e-6 * abs(float(want.detach())) + 1e-7
    (want * 0.5).backward()
    (got * 0.5).backward()
    ga, gb = xa.grad.cpu().numpy(), xb.grad.cpu().numpy()
    assert np.abs(gb - ga).max() <= 2e-6 * np.abs(ga).max() + 1e-9
    if R > 2:
        assert not gb[2::3].any()
    # the unit seed of backward_losses passes the stored gradient through untouched
    from pet.lib.ops import _hip as H
    xc = x.clone().requires_grad_(True)
    torch.autograd.backward([ops.cross_entropy_fused(xc, lab)], [H.unit_seed(x.device)])
    np.testing.assert_array_equal(xc.grad.cpu().numpy() * np.float32(0.5), gb)


def test_cross_entropy_fused_with_no_valid_row_is_the_frameworks_nan_and_zero_gradient():
    import pet.lib.ops as ops
    x = torch.randn(6, 81).cuda().requires_grad_(True)
    lab = torch.full((6,), -100, dtype=torch.int64).cuda()
    got = ops.cross_entropy_fused(x, lab)
    got.backward()
    assert torch.isnan(got) and not x.grad.any()
