"""GPU vs the CPU oracle at FULL size (BASELINE config #2: R-50-FPN CPM R-CNN, 2 x 3 x 800 x 1333 -> a 2 x 3 x 800 x
1344 batch), VERDICT r3 "missing 5": the dense path of pet/rcnn/modeling/model_builder.py:71-159 on the same weights --

  * C2..C5, P2..P6, the RPN logits / deltas of all five levels, and the three RoI heads (cls, the three grid stages
    incl. the ISM branch, RSM) on a fixed RoI list: every tensor within 1e-3 of its maximum -- end to end in exact
    fp32, stage by stage on the oracle's inputs in bf16x3 (and why: see that test);
  * FPN level indices of the REAL proposal set (the training-mode RPN's output on this batch) bit-equal to the
    oracle's LevelMapper;
  * NMS keep lists bit-equal to the oracle's greedy NMS on the real pre-NMS candidates of every (image, level): the
    top-2000 sigmoid scores of the GPU's own logits, decoded and clipped by the oracle;
  * the TEST-MODE forward (BASELINE config #1's workload: one 3 x 800 x 1333 image per forward) -- RPN test
    post-processing, cls head, CLSPostProcessor + ml_nms, three grid stages, ISM, RSM -- against
    oracle/cpu_pipeline.infer_image on the same weights: detections one to one (VERDICT r4 missing 2).

The oracle (oracle/cpu_model.py: torch-CPU fp32 convs + the C RoIAlign, pinned to the reference as its header says)
runs once per session: ~10 s on the GPU box's host cores."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last
H_IMG, W_IMG = 800, 1333


def rel(a, b):
    a = a.detach().float().cpu().contiguous()
    b = b.detach().float().cpu().contiguous() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b)).float()
    assert tuple(a.shape) == tuple(b.shape), (tuple(a.shape), tuple(b.shape))
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(scope="module")
def setup():
    import __graft_entry__ as entry
    entry.ensure_built()
    from bench import Trainer, calibrate_frozen_affine, synthetic_batch
    from oracle import cpu_model as M
    from pet.lib.ops import _hip
    from pet.rcnn.core import config
    device = torch.device("cuda", 0)
    prev = _hip.get_conv_math()
    _hip.set_conv_math("f32")
    tr = Trainer(device)
    images, targets = synthetic_batch(2, H_IMG, W_IMG, 16, 1234, device)
    cal, _ = synthetic_batch(2, H_IMG, W_IMG, 1, 4321, device)
    calibrate_frozen_affine(tr.model, cal.tensors)
    assert tuple(images.tensors.shape) == (2, 3, 800, 1344)
    sd = {}
    for k, v in tr.model.state_dict().items():
        if "cell_anchors" in k:
            continue
        t = v.detach().float().cpu().contiguous().clone()
        if t.dim() == 4 and (k.endswith("fc6.weight") or k.endswith("iou_fc1.weight")):
            t = t.reshape(t.shape[0], -1)               # the reference's [K, C*7*7] layout (ops.Linear(window=...))
        sd[k] = t
    # fixed RoI lists: boxes of every FPN level's size range, both images
    gen = torch.Generator().manual_seed(5)

    def rois(k):
        wh = torch.exp(torch.rand(k, 2, generator=gen) * (np.log(700.) - np.log(12.)) + np.log(12.))
        xy = torch.rand(k, 2, generator=gen) * torch.tensor([W_IMG - 16., H_IMG - 16.])
        x2 = torch.min(xy[:, 0] + wh[:, 0], torch.tensor(W_IMG - 1.))
        y2 = torch.min(xy[:, 1] + wh[:, 1], torch.tensor(H_IMG - 1.))
        img = (torch.arange(k) % 2).float()
        r = torch.stack([img, xy[:, 0], xy[:, 1], x2, y2], 1)
        return r[torch.argsort(img, stable=True)]     # image-contiguous, like the per-image BoxLists
    r_cls, r_grid = rois(256), rois(48)
    torch.set_num_threads(16)
    with torch.no_grad():
        x = images.tensors.cpu()
        c = M.backbone(sd, x)
        p = M.fpn(sd, c)
        lo, br = M.rpn_head(sd, p)
        want = {"c": c, "p": p, "lo": lo, "br": br,
                "cls": M.cls_head(sd, p, r_cls), "rsm": M.cls_head(sd, p, r_cls, "Head_rescore", "Output_rescore"),
                "grid": [M.grid_stage(sd, p, r_grid, s, last=(s == 2)) for s in range(3)]}
    yield dict(tr=tr, images=images, targets=targets, want=want, r_cls=r_cls, r_grid=r_grid, sd=sd)
    _hip.set_conv_math(prev)
    config.reset_cfg()
    torch.cuda.empty_cache()


def _boxlists(r, device):
    from pet.utils.data.structures.bounding_box import BoxList
    out = []
    for i in range(2):
        out.append(BoxList(r[r[:, 0] == i][:, 1:].to(device), (W_IMG, H_IMG), mode="xyxy"))
    return out


def _heads(G, p, b_cls, b_grid, want, worst, tag=""):
    worst[tag + "cls_logits"] = rel(G.Output_cls(G.Head_cls(p, b_cls)), want["cls"])
    worst[tag + "rescore_logits"] = rel(G.Output_rescore(G.Head_rescore(p, b_cls)), want["rsm"])
    for s in range(3):
        xg, _ = getattr(G, "Head_grid_%d" % s)(p, b_grid)
        hm, iou = getattr(G, "Output_grid_%d" % s)(xg, None)
        wx, wh, wi = want["grid"][s]
        worst[tag + "grid_feat_%d" % s] = rel(xg, wx)
        worst[tag + "grid_heat_%d" % s] = rel(hm["unfused"], wh)
        if s == 2:
            worst[tag + "ism_logits"] = rel(iou, wi)


def _end_to_end(S, math):
    from pet.lib.ops import _hip
    want = S["want"]
    model = S["tr"].model
    _hip.set_conv_math(math)
    worst = {}
    with torch.no_grad():
        c = model.Conv_Body(S["images"].tensors)
        p = model.Conv_Body_FPN(c)
        lo, br = model.RPN.head(p)
        for i in range(4):
            worst["C%d" % (i + 2)] = rel(c[i], want["c"][i])
        for i in range(5):
            worst["P%d" % (i + 2)] = rel(p[i], want["p"][i])
            worst["rpn_logits_%d" % i] = rel(lo[i], want["lo"][i])
            worst["rpn_deltas_%d" % i] = rel(br[i], want["br"][i])
        dev = p[0].device
        _heads(model.Grid_Cascade_RCNN, p, _boxlists(S["r_cls"], dev), _boxlists(S["r_grid"], dev), want, worst)
    return worst


def test_dense_path_at_full_size_matches_the_oracle_in_exact_fp32(setup):
    """north_star's bar as it stands: every tensor of the dense path within 1e-3 of its maximum, end to end, image to
    head outputs (measured: 1e-4 -- the summation-order noise of two fp32 implementations, ~1e-7 per layer, grown by
    this randomly initialised network, see the bf16x3 case below)."""
    worst = _end_to_end(setup, "f32")
    print("full-size oracle [f32, end to end]: worst %.2e (%s)" % (max(worst.values()), max(worst, key=worst.get)))
    bad = {k: v for k, v in worst.items() if not v < 1e-3}
    assert not bad, bad


def test_dense_path_at_full_size_matches_the_oracle_in_bf16x3(setup):
    """The headline arithmetic.  A bf16x3 layer is 5-7e-6 of its output's maximum away from fp32 (held at 1e-4 per
    layer in test_gpu_conv.py).  The benchmark network -- reference initialisers, every frozen affine calibrated to unit
    variance, so that each bottleneck adds a residual branch as large as its trunk -- GROWS any perturbation by ~1.26x
    per block (tools/err_growth.py; profiles/round4_error_growth.txt: rms error 2.8e-6 behind the stem, 1.8e-3 behind
    the 16th block, while every block's own error on exact inputs stays 5e-6 .. 1.6e-5), so end to end this arithmetic
    arrives at ~2e-3 at C5 on THIS network where exact fp32 arrives at 1e-4 from 1e-7.  Held here:
      * stage by stage on the ORACLE's inputs (stem + layer1 from the image, layer2 / layer3 / layer4 from the oracle's
        C2 / C3 / C4, the FPN from its C2..C5, the RPN head and the three RoI heads from its P2..P6): 1e-3, north_star's
        bar, for every tensor -- what the arithmetic itself is responsible for;
      * end to end: bounded at 5e-3 and printed (not north_star's bar: the growth above is the network's)."""
    from pet.lib.ops import _hip
    S, want = setup, setup["want"]
    model = S["tr"].model
    body = model.Conv_Body
    dev = S["images"].tensors.device
    _hip.set_conv_math("bf16x3")
    up = lambda t: t.to(dev).contiguous(memory_format=CL)
    worst = {}
    with torch.no_grad():
        feats = model.Conv_Body(S["images"].tensors)
        worst["stem+layer1 (from the image)"] = rel(feats[0], want["c"][0])
        for li in (2, 3, 4):
            x = up(want["c"][li - 2])
            for blk in getattr(body, "layer%d" % li):
                x = blk(x)
            worst["layer%d (from the oracle's C%d)" % (li, li)] = rel(x, want["c"][li - 1])
        p = model.Conv_Body_FPN([up(t) for t in want["c"]])
        for i in range(5):
            worst["P%d (from the oracle's C2..C5)" % (i + 2)] = rel(p[i], want["p"][i])
        po = [up(t) for t in want["p"]]
        lo, br = model.RPN.head(po)
        for i in range(5):
            worst["rpn_logits_%d" % i] = rel(lo[i], want["lo"][i])
            worst["rpn_deltas_%d" % i] = rel(br[i], want["br"][i])
        _heads(model.Grid_Cascade_RCNN, po, _boxlists(S["r_cls"], dev), _boxlists(S["r_grid"], dev), want, worst)
    print("full-size oracle [bf16x3, stage by stage on the oracle's inputs]: worst %.2e (%s)"
          % (max(worst.values()), max(worst, key=worst.get)))
    bad = {k: v for k, v in worst.items() if not v < 1e-3}
    assert not bad, bad
    e2e = _end_to_end(S, "bf16x3")
    print("full-size oracle [bf16x3, end to end]: worst %.2e (%s); C5 %.2e" % (max(e2e.values()), max(e2e, key=e2e.get),
                                                                               e2e["C5"]))
    bad = {k: v for k, v in e2e.items() if not v < 5e-3}
    assert not bad, bad


def test_level_indices_and_nms_keep_lists_on_the_real_proposal_set(setup):
    """bit-exact pieces (north_star: RoI indices / NMS keep masks) on what the full-size training step really
    produces: the RPN's proposals of this batch, and the pre-NMS candidates behind them."""
    import pet.lib.ops as ops
    from oracle import cpu_model as M
    from oracle import pyoracle as O
    from pet.lib.ops import _hip
    S = setup
    model = S["tr"].model
    _hip.set_conv_math("bf16x3")
    model.train()
    with torch.no_grad():
        feats = model._features(S["images"].tensors)
        proposals, _ = model.RPN(S["images"], feats, S["targets"])
        boxlists = proposals.to_boxlists() if hasattr(proposals, "to_boxlists") else proposals
        boxes = torch.cat([b.bbox for b in boxlists]).float()
        img = torch.cat([torch.full((len(b),), float(i)) for i, b in enumerate(boxlists)]).to(boxes.device)
        assert boxes.shape[0] >= 1000, "the full-size batch must give a real proposal set, got %d" % boxes.shape[0]
        rois5 = torch.cat([img[:, None], boxes], 1)
        _, levels = ops.roi_align_fpn(feats[:4], rois5, (7, 7), M.SCALES, 2, return_levels=True)
        want_lv = O.level_map(boxes.cpu().numpy(), 2, 5)
        assert np.array_equal(levels.cpu().numpy().astype(np.int64), np.asarray(want_lv).astype(np.int64))
        assert len(np.unique(want_lv)) >= 3, "proposals should spread over the FPN levels"
        # ---- NMS on the real candidates: per (image, level) the top-2000 scores of the GPU's own logits ------------
        lo, br = model.RPN.head(feats)
        segs_b, segs_s = [], []
        A = lo[0].shape[1]
        for l, stride in enumerate((4, 8, 16, 32, 64)):
            hl, wl = lo[l].shape[2], lo[l].shape[3]
            cell = O.cell_anchors(stride, (32 * 2 ** l,), (0.5, 1.0, 2.0))
            anchors = O.grid_anchors((hl, wl), stride, cell).astype(np.float32)          # [H*W*A, 4]
            for i in range(2):
                logit = lo[l][i].permute(1, 2, 0).reshape(-1)                           # (h, w, a) order
                delta = br[l][i].reshape(A, 4, hl, wl).permute(2, 3, 0, 1).reshape(-1, 4)
                k = min(2000, logit.numel())
                sc, idx = torch.sigmoid(logit).topk(k)
                # candidates with a score another candidate also has are dropped: the order among equal scores is
                # open in the reference (torchvision leaves it unspecified), everything else is exact
                u, first = np.unique(sc.cpu().numpy(), return_index=True)
                first = np.sort(first[np.argsort(-u, kind="stable")])
                sel = torch.from_numpy(first).to(sc.device)
                sc, idx = sc[sel], idx[sel]
                order = torch.argsort(sc, descending=True, stable=True)
                sc, idx = sc[order], idx[order]
                dec = O.box_decode(delta[idx].cpu().numpy(), anchors[idx.cpu().numpy()])
                dec[:, 0::2] = np.clip(dec[:, 0::2], 0, W_IMG - 1)
                dec[:, 1::2] = np.clip(dec[:, 1::2], 0, H_IMG - 1)
                segs_b.append(dec.astype(np.float32))
                segs_s.append(sc.cpu().numpy().astype(np.float32))
        offs = np.concatenate([[0], np.cumsum([len(s) for s in segs_s])]).tolist()
        allb = torch.from_numpy(np.concatenate(segs_b)).to(boxes.device)
        alls = torch.from_numpy(np.concatenate(segs_s)).to(boxes.device)
        keep, counts = ops.nms_segments(allb, alls, None, offs, 0.7, 0)
        keep, counts = keep.cpu().numpy(), counts.cpu().numpy()
        total_kept = 0
        for p_, (b, s) in enumerate(zip(segs_b, segs_s)):
            assert len(np.unique(s)) == len(s)
            want_keep = np.asarray(O.nms(b, s, 0.7)).astype(np.int64)
            got = keep[offs[p_]:offs[p_] + counts[p_]].astype(np.int64)
            assert np.array_equal(got, want_keep), "segment %d: %d vs %d kept" % (p_, len(got), len(want_keep))
            total_kept += len(got)
        assert total_kept > 2000


def _match_detections(gb, gs, gl, ob, os_, ol, extent):
    """one-to-one matching of two detection sets: same label, nearest box; -> (pairs, box deviation per pair)"""
    used, pairs = set(), []
    for i in range(len(gb)):
        cand = [j for j in np.flatnonzero(ol == gl[i]) if j not in used]
        if not cand:
            continue
        d = np.abs(ob[cand] - gb[i]).max(axis=1)
        j = cand[int(np.argmin(d))]
        # a box moved by arg-max flips -- one of a side's three voting points moved by one cell of a stage's window --
        # stays within ~ (1 + ratio) * side / 56 per stage of its twin; anything further is another detection
        side = max(ob[j, 2] - ob[j, 0], ob[j, 3] - ob[j, 1])
        if d.min() <= 0.15 * side + 1.0:
            used.add(j)
            pairs.append((i, j, float(d.min())))
    return pairs


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_test_mode_forward_at_full_size_matches_the_oracle(setup, math):
    """BASELINE config #1's workload on the HIP path: Generalized_RCNN.forward in eval mode on ONE 3 x 800 x 1344 image
    per forward (the reference's inference is per image: TEST.IMS_PER_GPU = 1) against
    oracle/cpu_pipeline.infer_image on the same weights -- grid_cascade_rcnn.py:92-100,161-224 and inference.py:59-124,
    145-298 of the reference: RPN test post-processing (1000 / 1000 / 1000), cls head, softmax, score > t & label != 0,
    multi-label NMS 0.3, three grid stages refining the kept boxes, ISM (score x IoU logit), RSM (s^0.8 p^0.2).

    The benchmark network's class scores are all ~1/81 (cls_score is initialised with std 0.01), so the reference's
    threshold 0.03 would pass nothing: the test places the threshold inside the WIDEST gap of the GPU's own sorted
    foreground scores around rank ~200, the same value for both sides -- candidates are then decided by a margin, not a
    tie.  What is held:
      * exact f32: the two detection sets pair off one to one (same label, same candidate) -- every detection except
        a counted handful (<= 2 %: a proposal at an NMS / top-k tie decided differently by two fp32 sums); paired
        boxes within 1e-3 of the image extent (1.3 px; measured ~1e-3 px), except counted arg-max flips of a heat map
        whose two best cells tie to rounding (<= 2 %, bounded by a cell's size); scores within 1e-3 relative (NaN where
        the reference's own s ** 0.8 is NaN: a negative ISM logit);
      * bf16x3: the same pairing with the bars the arithmetic's 2e-3 end-to-end feature error on THIS network allows
        (DESIGN section 5): >= 80 % of the detections paired, the paired ones judged as above with a 10 % flip budget;
        printed."""
    from oracle import cpu_pipeline as P
    from pet.lib.ops import _hip
    S = setup
    model = S["tr"].model
    G = model.Grid_Cascade_RCNN
    _hip.set_conv_math(math)
    model.eval()
    post = G.cls_post_processor
    saved = post.score_thresh
    extent = float(max(H_IMG, W_IMG))
    torch.set_num_threads(16)
    try:
        summary = []
        for i in range(2):
            x = S["images"].tensors[i:i + 1]
            with torch.no_grad():
                # the GPU's own class probabilities, to place the threshold
                feats = model._features(x)
                from pet.utils.data.structures.image_list import to_image_list
                props, _ = model.RPN(to_image_list(x), feats, None)
                assert len(props) == 1 and 500 <= len(props[0]) <= 1000, len(props[0])
                prob = torch.softmax(G.Output_cls(G.Head_cls(feats, props)), -1)[:, 1:].reshape(-1)
                top = torch.sort(prob, descending=True)[0][:400].cpu().numpy().astype(np.float64)
                gaps = top[120:300] - top[121:301]
                k = 120 + int(np.argmax(gaps))
                thr = float(0.5 * (top[k] + top[k + 1]))
                assert gaps.max() > 1e-7 * top[k], "no usable gap among the class scores"
                post.score_thresh = thr
                res = model(x)
            assert len(res) == 1
            r = res[0]
            gb = r.bbox.cpu().numpy()
            gs = r.get_field("scores").cpu().numpy()
            gl = r.get_field("labels").cpu().numpy()
            ob, os_, ol = P.infer_image(S["sd"], x.cpu().contiguous(), score_thresh=thr)
            assert len(ob) >= 30 and len(gb) >= 30, (len(gb), len(ob))
            assert (gl > 0).all() and (gl < 81).all()
            pairs = _match_detections(gb, gs, gl, ob, os_, ol, extent)
            n = max(len(gb), len(ob))
            unpaired = n - len(pairs)
            dev = np.array([d for _, _, d in pairs])
            moved = dev > 1e-3 * extent
            sc_bad = 0
            for (a, b_, d) in pairs:
                if d > 1e-3 * extent:
                    continue
                if np.isnan(os_[b_]) or np.isnan(gs[a]):
                    sc_bad += int(np.isnan(os_[b_]) != np.isnan(gs[a]))
                else:
                    sc_bad += int(abs(gs[a] - os_[b_]) > (1e-3 if math == "f32" else 5e-3) * abs(os_[b_]) + 1e-9)
            summary.append((i, len(gb), len(ob), unpaired, int(moved.sum()), sc_bad, thr))
            if math == "f32":
                assert unpaired <= max(1, int(0.02 * n)), summary[-1]
                assert int(moved.sum()) <= max(1, int(0.02 * n)), summary[-1]
                assert sc_bad <= max(1, int(0.02 * n)), summary[-1]
            else:
                assert unpaired <= int(0.2 * n), summary[-1]
                assert int(moved.sum()) <= max(2, int(0.1 * n)), summary[-1]
                assert sc_bad <= max(2, int(0.1 * n)), summary[-1]
        import os
        from conftest import ROOT
        line = "test_mode_fullsize[%s] per image (gpu dets, oracle dets, unpaired, moved boxes, score mismatches, thr): %s" \
            % (math, "; ".join("img%d %d %d %d %d %d %.6f" % t for t in summary))
        print(line)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_log.txt"), "a") as f:
            f.write(line + "\n")
    finally:
        post.score_thresh = saved
        model.train()
