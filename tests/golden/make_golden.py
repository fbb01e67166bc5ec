"""Generate the committed golden vectors by running the REFERENCE here.

Runs only in the build container (needs /root/reference and oracle/_ref).  The
GPU box never runs this; it only reads the .npz/.json files it wrote.

How the reference is driven (nothing from it is copied into the repo):
  * RoIAlign: oracle/_ref = the reference's own ROIAlign_cpu.cpp compiled by
    oracle/build_ref.py.
  * Python modules are imported from /root/reference.  Third-party packages the
    image lacks are replaced IN THIS PROCESS ONLY by inert stand-ins:
      apex.amp.float_function -> identity decorator
      torchvision.ops.nms / cv2 / pycocotools -> placeholders that raise if called
      pet.lib.ops._C -> the compiled reference RoIAlign; every other symbol raises
    np.float (removed in numpy 2) is aliased to float, torch.Tensor.cuda is bound
    to identity while the grid target/decoder run (they hard-code .cuda()).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

from detfill import det_fill_, rpn_head_feature, soften_heatmaps_  # noqa: E402
from oracle.build_ref import build as build_ref  # noqa: E402


def install_standins(ref_ext):
    np.float = float  # noqa: anchor_generator.py:230

    apex = types.ModuleType("apex")
    amp = types.ModuleType("apex.amp")
    amp.float_function = lambda f: f
    apex.amp = amp
    sys.modules["apex"] = apex
    sys.modules["apex.amp"] = amp

    def _absent(*a, **k):
        raise RuntimeError("third-party op absent in this container")

    tv = types.ModuleType("torchvision")
    tvo = types.ModuleType("torchvision.ops")
    tvo.nms = _absent
    tv.ops = tvo
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tvo
    for name in ("cv2", "pycocotools", "pycocotools.mask", "pycocotools.coco", "pycocotools.cocoeval"):
        sys.modules[name] = types.ModuleType(name)

    class _C(types.ModuleType):
        def __getattr__(self, item):
            return _absent
    c = _C("pet.lib.ops._C")
    c.roi_align_forward = ref_ext.roi_align_forward
    c.roi_align_backward = ref_ext.roi_align_backward
    sys.modules["pet.lib.ops._C"] = c
    sys.path.insert(0, REF)


def load_cfg(yaml_rel):
    import yaml
    from pet.rcnn.core import config
    from pet.utils.collections import AttrDict
    with open(os.path.join(REF, yaml_rel)) as f:
        y = AttrDict(yaml.safe_load(f))
    config._merge_a_into_b(y, config.cfg)
    return config.cfg


def gen_ops(ref_ext, out):
    rng = np.random.default_rng(20240)
    # ---- RoIAlign: (7|14) x scales, edge RoIs ------------------------------------------------
    for ph, scale, H, W in [(7, 0.25, 24, 40), (14, 0.25, 24, 40), (7, 0.125, 12, 20), (14, 0.125, 12, 20),
                            (7, 1 / 16., 6, 10), (14, 1 / 16., 6, 10), (7, 1 / 32., 3, 5), (14, 1 / 32., 3, 5)]:
        C, B, K = 6, 2, 24
        x = rng.standard_normal((B, C, H, W)).astype(np.float32)
        b = rng.integers(0, B, K).astype(np.float32)
        iw, ih = W / scale, H / scale
        x1 = rng.uniform(-0.2 * iw, iw, K)
        y1 = rng.uniform(-0.2 * ih, ih, K)
        w = rng.uniform(0.05, 0.9 * iw, K)
        h = rng.uniform(0.05, 0.9 * ih, K)
        rois = np.stack([b, x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
        # hand-made edge cases: outside the map, sub-pixel, integer aligned, border straddling
        rois[0] = [0, -500, -500, -400, -400]
        rois[1] = [1, iw + 50, ih + 50, iw + 90, ih + 120]
        rois[2] = [0, 10.25, 7.5, 10.5, 7.75]
        rois[3] = [1, 0, 0, 4 / scale, 4 / scale]
        rois[4] = [0, iw - 9, ih - 9, iw + 30, ih + 30]
        rois[5] = [1, -13, 5, 17, 31]
        key = "roi_%d_%d" % (ph, int(round(1 / scale)))
        for interp in (0, 1):
            for aligned in (False, True):
                y = ref_ext.roi_align_forward(torch.from_numpy(x), torch.from_numpy(rois), scale, ph, ph, 2,
                                              aligned, interp)
                g = rng.standard_normal(tuple(y.shape)).astype(np.float32)
                gi = ref_ext.roi_align_backward(torch.from_numpy(g), torch.from_numpy(rois), scale, ph, ph, B, C, H,
                                                W, 2, aligned, interp)
                sfx = "_i%d_a%d" % (interp, int(aligned))
                out[key + sfx + "_y"] = y.numpy()
                out[key + sfx + "_g"] = g
                out[key + sfx + "_gi"] = gi.numpy()
        out[key + "_x"] = x
        out[key + "_rois"] = rois
    # adaptive sampling ratio (sampling_ratio=0)
    x = rng.standard_normal((1, 3, 16, 16)).astype(np.float32)
    rois = np.array([[0, 3, 2, 40, 50], [0, 0, 0, 63, 63], [0, 20, 20, 22, 21]], np.float32)
    out["roi_adapt_x"], out["roi_adapt_rois"] = x, rois
    out["roi_adapt_y"] = ref_ext.roi_align_forward(torch.from_numpy(x), torch.from_numpy(rois), 0.25, 7, 7, 0,
                                                   False, 0).numpy()

    # ---- LevelMapper (poolers.py:9-40) -------------------------------------------------------
    from pet.rcnn.utils.poolers import LevelMapper
    from pet.utils.data.structures.bounding_box import BoxList
    n = 4096
    x1 = rng.uniform(0, 1000, n)
    y1 = rng.uniform(0, 600, n)
    w = np.exp(rng.uniform(np.log(2), np.log(1200), n))
    h = np.exp(rng.uniform(np.log(2), np.log(800), n))
    boxes = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    k = 0
    for s in (56, 112, 224, 448, 896):       # exact level boundaries: area+1 = s^2
        for d in (-1, 0, 1):
            boxes[k] = [10, 10, 10 + s - 1 + d, 10 + s - 1]
            k += 1
    lm = LevelMapper(2, 5)
    out["lvl_boxes"] = boxes
    out["lvl_out"] = lm([BoxList(torch.from_numpy(boxes), (1333, 800))]).numpy()

    # ---- anchors (anchor_generator.py) -------------------------------------------------------
    from pet.rcnn.modeling.rpn.anchor_generator import generate_anchors, AnchorGenerator
    out["anchors_matlab_table"] = generate_anchors(16, (128, 256, 512), (0.5, 1, 2)).float().numpy()
    ag = AnchorGenerator(sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0),
                         anchor_strides=(4, 8, 16, 32, 64), straddle_thresh=0)
    for i, ca in enumerate(ag.cell_anchors):
        out["cell_anchors_%d" % i] = ca.numpy()
    grids = ag.grid_anchors([(5, 7), (3, 4), (2, 2), (1, 2), (1, 1)])
    for i, g in enumerate(grids):
        out["grid_anchors_%d" % i] = g.numpy()
    bl = BoxList(grids[0], (28, 20), mode="xyxy")
    ag.add_visibility_to(bl)
    out["grid_anchors_0_visibility"] = bl.get_field("visibility").numpy()

    # ---- BoxCoder ----------------------------------------------------------------------------
    from pet.rcnn.utils.box_coder import BoxCoder
    bc = BoxCoder((1., 1., 1., 1.))
    n = 512
    pb = boxes[:n].copy()
    codes = (rng.standard_normal((n, 4)) * np.array([0.3, 0.3, 1.5, 1.5])).astype(np.float32)
    codes[:8, 2:] = 6.0  # exercises the log(1000/16) clip
    out["bc_boxes"], out["bc_codes"] = pb, codes
    out["bc_decode"] = bc.decode(torch.from_numpy(codes), torch.from_numpy(pb)).numpy()
    gt = boxes[n:2 * n].copy()
    out["bc_gt"] = gt
    out["bc_encode"] = bc.encode(torch.from_numpy(gt), torch.from_numpy(pb)).numpy()

    # ---- boxlist_iou (+1) and Matcher --------------------------------------------------------
    from pet.utils.data.structures.boxlist_ops import boxlist_iou
    from pet.rcnn.utils.matcher import Matcher
    g16 = boxes[100:116]
    # proposals that overlap the gts: jittered copies + randoms + exact copies (ties)
    props = np.concatenate([g16 + rng.uniform(-15, 15, g16.shape).astype(np.float32), boxes[200:400], g16,
                            g16[:4]], 0).astype(np.float32)
    iou = boxlist_iou(BoxList(torch.from_numpy(g16), (1333, 800)), BoxList(torch.from_numpy(props), (1333, 800)))
    out["iou_gt"], out["iou_props"], out["iou_out"] = g16, props, iou.numpy()
    out["match_rpn"] = Matcher(0.7, 0.3, allow_low_quality_matches=True)(iou.clone()).numpy()
    out["match_cls"] = Matcher(0.5, 0.5, allow_low_quality_matches=False)(iou.clone()).numpy()
    out["match_g2"] = Matcher(0.7, 0.7, allow_low_quality_matches=False)(iou.clone()).numpy()

    # ---- losses ------------------------------------------------------------------------------
    from pet.lib.ops import smooth_l1_loss, l2_loss
    a = rng.standard_normal((64, 4)).astype(np.float32)
    b = rng.standard_normal((64, 4)).astype(np.float32) * 0.3
    out["sl1_a"], out["sl1_b"] = a, b
    out["sl1_out"] = smooth_l1_loss(torch.from_numpy(a), torch.from_numpy(b), beta=1. / 9, reduction="sum").numpy()
    t = np.abs(rng.standard_normal((32, 2))).astype(np.float32)
    t[::3] = 0
    lg = rng.standard_normal((32, 2)).astype(np.float32)
    out["l2_x"], out["l2_t"] = lg, t
    out["l2_out"] = l2_loss(torch.from_numpy(lg), torch.from_numpy(t)).numpy()


def gen_grid(out):
    """GridLossComputation.prepare_target and GridPostProcessor.get_boxes (per stage ratio)."""
    from pet.rcnn.modeling.grid_cascade_rcnn.loss import loss_evaluator, calc_sub_regions
    from pet.rcnn.modeling.grid_cascade_rcnn.inference import post_processor
    from pet.utils.data.structures.bounding_box import BoxList
    rng = np.random.default_rng(777)
    out["sub_regions"] = np.array(calc_sub_regions(9, 3, 56), np.int32)
    R = 20
    x1 = rng.uniform(0, 900, R)
    y1 = rng.uniform(0, 500, R)
    w = rng.uniform(8, 400, R)
    h = rng.uniform(8, 300, R)
    gt = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    jit = rng.uniform(-0.2, 0.2, (R, 4)) * np.stack([w, h, w, h], 1)
    boxes = (gt + jit).astype(np.float32)
    boxes[0] = [100, 100, 102, 150]        # width <= grid_size -> skipped (loss.py:215-217)
    boxes[1] = gt[1]                        # RoI == gt
    gt[2] = [boxes[2][0] - 300, boxes[2][1] - 200, boxes[2][0] - 10, boxes[2][1] - 5]   # points fall outside
    out["grid_boxes"], out["grid_gt"] = boxes, gt
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    orig_get_device = torch.Tensor.get_device
    try:
        for stage in range(3):
            ev = loss_evaluator(stage=stage, type="grid")
            ev.pos_result = (torch.from_numpy(boxes.copy()), torch.from_numpy(gt.copy()))
            out["grid_targets_s%d" % stage] = ev.prepare_target(None, None).numpy()
            pp = post_processor(stage=stage, type="grid")
            logits = (rng.standard_normal((R, 9, 28, 28)) * 2).astype(np.float32)
            logits[3, 4] = 0.0                        # a flat map: argmax tie -> first index
            logits[5, :, 10, 11] = 25.0               # saturating sigmoid
            bl = BoxList(torch.from_numpy(boxes.copy()), (1333, 800))
            res = pp.get_boxes(bl, torch.from_numpy(logits), False)
            out["grid_logits_s%d" % stage] = logits
            out["grid_decode_s%d" % stage] = res.numpy()
    finally:
        torch.Tensor.cuda = saved_cuda
        torch.Tensor.get_device = orig_get_device


def gen_model(out, meta):
    """Structural golden: the reference module tree with name-keyed deterministic weights."""
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    from pet.utils.data.structures.bounding_box import BoxList
    from pet.utils.data.structures.image_list import to_image_list
    torch.manual_seed(0)
    model = Generalized_RCNN(is_train=True)
    model = convert_bn2affine_model(model, merge=True)
    meta["state_dict"] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    meta["trainable"] = [k for k, p in model.named_parameters() if p.requires_grad]
    det_fill_(model)
    model.eval()
    rng = np.random.default_rng(5)
    img = (rng.uniform(0, 255, (1, 3, 64, 96)) - np.array([102.9801, 115.9465, 122.7717]).reshape(1, 3, 1, 1))
    img = img.astype(np.float32)
    out["m_img"] = img
    with torch.no_grad():
        x = torch.from_numpy(img)
        c = model.Conv_Body(x)
        for i, t in enumerate(c):
            out["m_c%d" % (i + 2)] = t.numpy()[:, ::8]           # every 8th channel
        p = model.Conv_Body_FPN(c)
        for i, t in enumerate(p):
            out["m_p%d" % (i + 2)] = t.numpy()[:, ::8]
        logits, breg = model.RPN.head(p)
        for i, (a, b) in enumerate(zip(logits, breg)):
            out["m_rpn_logits_%d" % i] = a.numpy()
            out["m_rpn_bbox_%d" % i] = b.numpy()
        rois = np.array([[2, 3, 60, 40], [10, 5, 90, 60], [0, 0, 95, 63], [30, 20, 50, 45], [5, 30, 25, 62],
                         [40, 2, 70, 20]], np.float32)
        out["m_rois"] = rois
        boxes = [BoxList(torch.from_numpy(rois), (96, 64))]
        g = model.Grid_Cascade_RCNN
        f = g.Head_cls(p, boxes)
        out["m_cls_feat"] = f.numpy()
        out["m_cls_logits"] = g.Output_cls(f).numpy()
        f = g.Head_rescore(p, boxes)
        out["m_rescore_logits"] = g.Output_rescore(f).numpy()
        for s in range(3):
            head = getattr(g, "Head_grid_%d" % s)
            outp = getattr(g, "Output_grid_%d" % s)
            outp.train()            # "unfused" branch is the training branch (outputs.py:66)
            xg, xso = head(p, boxes)
            hm, iou = outp(xg, xso)
            out["m_grid_feat_%d" % s] = xg.numpy()[:, ::16]
            out["m_grid_heat_%d" % s] = hm["unfused"].numpy()
            if iou is not None:
                out["m_grid_iou_%d" % s] = iou.numpy()
    # gradient golden for the grid stage 2 + cls head (small): d(sum of squares)/d(weights) checksums
    model.train()
    for q in model.parameters():
        if q.grad is not None:
            q.grad = None
    x = torch.from_numpy(img)
    c = model.Conv_Body(x)
    p = model.Conv_Body_FPN(c)
    xg, xso = g.Head_grid_2(p, boxes)
    hm, iou = g.Output_grid_2(xg, xso)
    loss = (hm["unfused"] ** 2).mean() + (iou ** 2).mean() + (g.Output_cls(g.Head_cls(p, boxes)) ** 2).mean()
    lo, br = model.RPN.head(p)
    loss = loss + sum((a ** 2).mean() for a in lo) + sum((a ** 2).mean() for a in br)
    loss.backward()
    out["m_loss"] = loss.detach().numpy()
    grads = {}
    for k, q in model.named_parameters():
        if q.grad is not None:
            gsum = q.grad.double()
            grads[k] = [float(gsum.sum()), float(gsum.abs().sum()), float((gsum ** 2).sum())]
    meta["grad_stats"] = grads
    for k in ("Conv_Body.layer2.0.conv1.weight", "Conv_Body.layer4.2.conv3.weight", "Conv_Body_FPN.fpn_out.2.weight",
              "RPN.head.conv.weight", "Grid_Cascade_RCNN.Head_grid_2.convs.0.0.weight",
              "Grid_Cascade_RCNN.Head_grid_2.convs.7.1.weight", "Grid_Cascade_RCNN.Output_grid_2.deconv_1.weight",
              "Grid_Cascade_RCNN.Output_grid_2.deconv_2.weight", "Grid_Cascade_RCNN.Output_grid_2.norm1.weight",
              "Grid_Cascade_RCNN.Output_grid_2.iou_pred.weight", "Grid_Cascade_RCNN.Output_cls.cls_score.weight"):
        gq = dict(model.named_parameters())[k].grad
        out["m_grad::" + k] = gq.numpy().reshape(-1)[::max(1, gq.numel() // 4096)]


def gen_x101_meta():
    """State-dict ABI of BASELINE config #5 (X-101-64x4d-FPN + DCN): names, shapes, trainable set.  Structure only:
    the reference's deformable conv is CUDA-only, so no forward can be run here."""
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    load_cfg("cfgs/rcnn/mscoco/grid_cascade/iou_helper/rescore/backbone/"
             "e2e_grid_cascade@567_rcnn_X-101b-64x4d-FPN-DCN_2x.yaml").DEVICE = "cpu"
    model = convert_bn2affine_model(Generalized_RCNN(is_train=True), merge=True)
    meta = {"state_dict": [[k, list(v.shape)] for k, v in model.state_dict().items()],
            "trainable": [k for k, p in model.named_parameters() if p.requires_grad]}
    with open(os.path.join(HERE, "model_x101_meta.json"), "w") as f:
        json.dump(meta, f)
    print("x101:", len(meta["state_dict"]), "keys,", len(meta["trainable"]), "trainable")


def gen_r101_meta():
    """State-dict ABI of BASELINE config #4 (R-101-FPN CPM R-CNN): names, shapes, trainable set."""
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    load_cfg("cfgs/rcnn/mscoco/grid_cascade/iou_helper/rescore/backbone/"
             "e2e_grid_cascade@567_rcnn_R-101-FPN_2x.yaml").DEVICE = "cpu"
    model = convert_bn2affine_model(Generalized_RCNN(is_train=True), merge=True)
    meta = {"state_dict": [[k, list(v.shape)] for k, v in model.state_dict().items()],
            "trainable": [k for k, p in model.named_parameters() if p.requires_grad]}
    with open(os.path.join(HERE, "model_r101_meta.json"), "w") as f:
        json.dump(meta, f)
    print("r101:", len(meta["state_dict"]), "keys,", len(meta["trainable"]), "trainable")


def gen_cascade():
    """Offset-regression Cascade R-CNN with ISM + RSM (cfgs/rcnn/mscoco/cascade/ISM+RSM): state-dict ABI,
    the RoI head in evaluation mode (decode / refine / ensemble / IoU-merged scores) and in training mode on a
    proposal set small enough that the 512-per-image sampler keeps everything (so the run is deterministic), all on
    the reference's own modules with name-keyed deterministic weights."""
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    from pet.utils.data.structures.bounding_box import BoxList
    cfg = load_cfg("cfgs/rcnn/mscoco/cascade/ISM+RSM/e2e_cascade_rcnn@2_R-50-FPN_1x.yaml")
    cfg.DEVICE = "cpu"
    torch.manual_seed(0)
    model = convert_bn2affine_model(Generalized_RCNN(is_train=True), merge=True)
    meta = {"state_dict": [[k, list(v.shape)] for k, v in model.state_dict().items()],
            "trainable": [k for k, p in model.named_parameters() if p.requires_grad]}
    det_fill_(model)
    out = {}
    rng = np.random.default_rng(5)
    img = (rng.uniform(0, 255, (1, 3, 64, 96)) - np.array([102.9801, 115.9465, 122.7717]).reshape(1, 3, 1, 1))
    img = img.astype(np.float32)
    out["img"] = img
    gt = np.array([[4, 6, 58, 44], [30, 10, 92, 60], [50, 30, 70, 50]], np.float32)
    gt_labels = np.array([3, 17, 80], np.int64)
    jit = np.array([[2, 3, 60, 40], [10, 5, 90, 60], [0, 0, 95, 63], [30, 20, 50, 45], [5, 30, 25, 62], [40, 2, 70, 20],
                    [28, 12, 90, 58], [5, 8, 55, 41], [52, 29, 71, 52], [48, 33, 68, 49], [60, 40, 90, 60],
                    [1, 1, 20, 20]], np.float32)
    out["gt"], out["gt_labels"], out["rois"] = gt, gt_labels, jit

    def props():
        b = BoxList(torch.from_numpy(np.concatenate([jit, gt]).copy()), (96, 64))
        b.add_field("objectness", torch.linspace(0.9, 0.1, len(b)))
        return [b]

    def targets():
        t = BoxList(torch.from_numpy(gt.copy()), (96, 64))
        t.add_field("labels", torch.from_numpy(gt_labels.copy()))
        return [t]
    head = model.Cascade_RCNN
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with torch.no_grad():
            p = model.Conv_Body_FPN(model.Conv_Body(torch.from_numpy(img)))
            model.eval()
            x, result, _ = head(p, props())
            out["eval_x"] = x.numpy()
            out["eval_bbox"] = result[0].bbox.numpy()
            out["eval_scores"] = result[0].get_field("scores").numpy()
            for s in (1, 2):
                f = getattr(head, "Box_Head_%d" % s)(p, props())
                c, b, i = getattr(head, "Output_%d" % s)(f)
                out["s%d_cls" % s], out["s%d_bbox" % s] = c.numpy(), b.numpy()
                if i is not None:
                    out["s%d_iou" % s] = i.numpy()
        model.train()
        for q in model.parameters():
            q.grad = None
        p = model.Conv_Body_FPN(model.Conv_Body(torch.from_numpy(img)))
        x, proposals, losses = head(p, props(), targets())
        for k, v in losses.items():
            out["loss::" + k] = np.asarray(float(v), np.float64)
        out["train_final_bbox"] = proposals[0].bbox.detach().numpy()
        out["train_final_labels"] = proposals[0].get_field("labels").numpy()
        sum(losses.values()).backward()
        grads = {}
        for k, q in model.named_parameters():
            if q.grad is not None and k.startswith("Cascade_RCNN"):
                g = q.grad.double()
                grads[k] = [float(g.sum()), float(g.abs().sum()), float((g ** 2).sum())]
        meta["grad_stats"] = grads
    finally:
        torch.Tensor.cuda = saved_cuda
    np.savez_compressed(os.path.join(HERE, "model_cascade.npz"), **out)
    with open(os.path.join(HERE, "model_cascade_meta.json"), "w") as f:
        json.dump(meta, f)
    print("cascade:", len(meta["state_dict"]), "keys;", {k: float(v) for k, v in out.items() if k.startswith("loss::")})


def _cpm_cfg():
    cfg = load_cfg("cfgs/rcnn/mscoco/grid_cascade/iou_helper/rescore/e2e_grid_cascade@567_rcnn_R-50-FPN_2x.yaml")
    cfg.DEVICE = "cpu"
    return cfg


def _det_image(rng, n, h, w):
    img = rng.uniform(0, 255, (n, 3, h, w)) - np.array([102.9801, 115.9465, 122.7717]).reshape(1, 3, 1, 1)
    return img.astype(np.float32)


def _grad_stats(model, prefix=""):
    grads = {}
    for k, q in model.named_parameters():
        if q.grad is not None and k.startswith(prefix):
            g = q.grad.double()
            grads[k] = [float(g.sum()), float(g.abs().sum()), float((g ** 2).sum())]
    return grads


def gen_model_big():
    """A second structural golden at a size where error accumulation through the 50-layer backbone, the FPN and the
    8-conv grid stacks is visible: 1 x 3 x 256 x 320 image, 64 RoIs spread over the four RoI levels.
    Forward tensors are stored sub-sampled (channel and spatial strides) to keep the fixture small; gradients as
    L1 / L2 statistics of all 196 trainable tensors plus strided samples of eleven of them."""
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    from pet.utils.data.structures.bounding_box import BoxList
    _cpm_cfg()
    torch.manual_seed(0)
    model = convert_bn2affine_model(Generalized_RCNN(is_train=True), merge=True)
    det_fill_(model)
    model.eval()
    out, meta = {}, {}
    rng = np.random.default_rng(11)
    H, W, R = 256, 320, 64
    img = _det_image(rng, 1, H, W)
    out["img"] = img
    side = np.exp(rng.uniform(np.log(12), np.log(300), R))
    asp = np.exp(rng.uniform(-0.7, 0.7, R))
    bw, bh = np.minimum(side * asp, W - 2), np.minimum(side / asp, H - 2)
    x1, y1 = rng.uniform(0, W - 1 - bw), rng.uniform(0, H - 1 - bh)
    rois = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
    out["rois"] = rois
    boxes = [BoxList(torch.from_numpy(rois), (W, H))]
    g = model.Grid_Cascade_RCNN
    with torch.no_grad():
        c = model.Conv_Body(torch.from_numpy(img))
        for i, t in enumerate(c):
            out["c%d" % (i + 2)] = t.numpy()[:, ::16, ::2, ::2]
        p = model.Conv_Body_FPN(c)
        for i, t in enumerate(p):
            out["p%d" % (i + 2)] = t.numpy()[:, ::16, ::2, ::2]
        logits, breg = model.RPN.head(p)
        for i, (a, b) in enumerate(zip(logits, breg)):
            out["rpn_logits_%d" % i] = a.numpy()[:, :, ::2, ::2]
            out["rpn_bbox_%d" % i] = b.numpy()[:, :, ::2, ::2]
        f = g.Head_cls(p, boxes)
        out["cls_feat"] = f.numpy()[:, ::4]
        out["cls_logits"] = g.Output_cls(f).numpy()
        out["rescore_logits"] = g.Output_rescore(g.Head_rescore(p, boxes)).numpy()
        for s in range(3):
            outp = getattr(g, "Output_grid_%d" % s)
            outp.train()
            xg, xso = getattr(g, "Head_grid_%d" % s)(p, boxes)
            hm, iou = outp(xg, xso)
            out["grid_feat_%d" % s] = xg.numpy()[:, ::16]
            out["grid_heat_%d" % s] = hm["unfused"].numpy()[:, :, ::2, ::2]
            if iou is not None:
                out["grid_iou_%d" % s] = iou.numpy()
    model.train()
    for q in model.parameters():
        q.grad = None
    p = model.Conv_Body_FPN(model.Conv_Body(torch.from_numpy(img)))
    loss = 0
    for s in range(3):
        xg, xso = getattr(g, "Head_grid_%d" % s)(p, boxes)
        hm, iou = getattr(g, "Output_grid_%d" % s)(xg, xso)
        loss = loss + (hm["unfused"] ** 2).mean()
        if iou is not None:
            loss = loss + (iou ** 2).mean()
    loss = loss + (g.Output_cls(g.Head_cls(p, boxes)) ** 2).mean()
    loss = loss + (g.Output_rescore(g.Head_rescore(p, boxes)) ** 2).mean()
    lo, br = model.RPN.head(p)
    loss = loss + sum((a ** 2).mean() for a in lo) + sum((a ** 2).mean() for a in br)
    loss.backward()
    out["loss"] = loss.detach().numpy()
    meta["grad_stats"] = _grad_stats(model)
    params = dict(model.named_parameters())
    for k in ("Conv_Body.layer2.0.conv1.weight", "Conv_Body.layer3.5.conv2.weight", "Conv_Body.layer4.2.conv3.weight",
              "Conv_Body_FPN.fpn_out.2.weight", "Conv_Body_FPN.fpn_in.2.weight", "RPN.head.conv.weight",
              "Grid_Cascade_RCNN.Head_grid_0.convs.0.0.weight", "Grid_Cascade_RCNN.Head_grid_2.convs.7.0.weight",
              "Grid_Cascade_RCNN.Head_grid_1.convs.3.1.weight", "Grid_Cascade_RCNN.Output_grid_2.deconv_1.weight",
              "Grid_Cascade_RCNN.Output_grid_2.iou_fc1.weight", "Grid_Cascade_RCNN.Head_cls.fc6.weight",
              "Grid_Cascade_RCNN.Output_rescore.cls_score.weight"):
        gq = params[k].grad
        out["grad::" + k] = gq.numpy().reshape(-1)[::max(1, gq.numel() // 4096)]
    np.savez_compressed(os.path.join(HERE, "model_r50_big.npz"), **out)
    with open(os.path.join(HERE, "model_r50_big_meta.json"), "w") as f:
        json.dump(meta, f)
    print("big:", len(out), "arrays, loss", float(loss), "grad tensors", len(meta["grad_stats"]))


def _jittered_proposals(rng, gt, n_per_gt, n_bg, W, H):
    props = []
    for b in gt:
        w, h = b[2] - b[0], b[3] - b[1]
        for j in range(n_per_gt):
            # a ladder of jitter amplitudes: IoUs from ~0.9 down to ~0.2 (positives and negatives of every stage)
            amp = 0.01 + 0.5 * (j / max(n_per_gt - 1, 1)) ** 1.5
            d = rng.uniform(-amp, amp, 4) * np.array([w, h, w, h])
            q = np.clip(b + d, [0, 0, 0, 0], [W - 1, H - 1, W - 1, H - 1])
            if q[2] - q[0] > 6 and q[3] - q[1] > 6:
                props.append(q)
    for _ in range(n_bg):
        w, h = rng.uniform(10, W / 2), rng.uniform(10, H / 2)
        x, y = rng.uniform(0, W - 1 - w), rng.uniform(0, H - 1 - h)
        props.append(np.array([x, y, x + w, y + h]))
    return np.asarray(props, np.float32)


def gen_cpm_train():
    """Rows a-11 / a-12: the REFERENCE GridCascadeRCNN.forward in training mode
    (grid_cascade_rcnn.py:57-224) on a 2-image batch whose proposal sets are small enough that every sampler keeps
    everything (cls sampler <= 128 positives / 384 negatives per image, <= 96 grid positives per image, KEEP_RATIO
    off), so the run involves no random draw: the 8 losses, the RoI set entering every grid stage, the boxes handed to
    the RSM head and gradient statistics.  Plus CLSPostProcessor's candidate selection (inference.py:59-124) with the
    CUDA-only ml_nms replaced by a recorder."""
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    from pet.utils.data.structures.bounding_box import BoxList
    import pet.rcnn.modeling.grid_cascade_rcnn.inference as ref_inf
    cfg = _cpm_cfg()
    assert not cfg.GRID_RCNN.RESCORE_OPTION.KEEP_RATIO and not cfg.GRID_RCNN.RANDOM_JITTER
    torch.manual_seed(0)
    model = convert_bn2affine_model(Generalized_RCNN(is_train=True), merge=True)
    det_fill_(model)
    soften_heatmaps_(model)
    from pet.utils.data.structures.boxlist_ops import boxlist_iou
    H, W = 160, 224
    gts = [np.array([[12, 20, 96, 130], [100, 8, 215, 90], [60, 70, 180, 150], [150, 100, 200, 155], [5, 5, 40, 44]],
                    np.float32),
           np.array([[30, 30, 190, 140], [8, 90, 70, 156], [120, 12, 160, 60]], np.float32)]
    gt_labels = [np.array([3, 17, 80, 1, 44], np.int64), np.array([9, 9, 62], np.int64)]
    head = model.Grid_Cascade_RCNN
    thr = cfg.GRID_RCNN.CASCADE_MAPPING_OPTION.FG_IOU_THRESHOLD

    def targets():
        res = []
        for g_, l_ in zip(gts, gt_labels):
            t = BoxList(torch.from_numpy(g_.copy()), (W, H))
            t.add_field("labels", torch.from_numpy(l_.copy()))
            res.append(t)
        return res

    # Recorders around the reference's own methods (nothing of the reference is modified): the RoI set entering each
    # stage, and the two margins that decide whether the fixture is robust against the 1e-5-level differences of
    # another conv arithmetic -- the distance of every decoded RoI's best IoU from the stage threshold it is matched
    # at, and the gap between the best and second-best cell of every decoded heat map.
    stage_rois, gaps, iou_margins = [], [], []
    orig_ftg = head._forward_train_grid

    def recording_ftg(stage, features, proposals_, targets=None):
        r = orig_ftg(stage, features, proposals_, targets=targets)
        stage_rois.append([b.bbox.detach().numpy().copy() for b in r[2]])
        return r
    restore = [(head, "_forward_train_grid", orig_ftg)]
    head._forward_train_grid = recording_ftg
    for s_ in range(3):
        ev = getattr(head, "grid_loss_evaluator_%d" % s_)
        orig_sub = ev.subsample

        def recording_sub(proposals_, targets_, _orig=orig_sub, _s=s_):
            if _s > 0:
                for p_, t_ in zip(proposals_, targets_):
                    q = boxlist_iou(t_, p_).max(dim=0)[0]
                    iou_margins.append(float((q - thr[_s]).abs().min()))
            return _orig(proposals_, targets_)
        ev.subsample = recording_sub
        restore.append((ev, "subsample", orig_sub))
    for s_ in range(2):
        pp = getattr(head, "grid_post_processor_%d" % s_)
        orig_gb = pp.get_boxes

        def recording_gb(proposals_, grid_pred, is_train, _orig=orig_gb):
            lg = grid_pred.detach().reshape(-1, grid_pred.shape[-1] * grid_pred.shape[-2])
            top = lg.topk(2, dim=1)[0]
            # best-vs-second-best cell, in units of the largest logit (the scale conv parity is measured in)
            gaps.append(((top[:, 0] - top[:, 1]).min() / lg.abs().max()).item())
            gaps.append(-float(lg.sigmoid().max()))     # (negated) largest probability: saturation check
            return _orig(proposals_, grid_pred, is_train)
        pp.get_boxes = recording_gb
        restore.append((pp, "get_boxes", orig_gb))
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        best = (-1.0, None)
        seeds = list(range(2024, 2064))
        for trial in range(len(seeds) + 1):
            seed = seeds[trial] if trial < len(seeds) else best[1]      # last pass: re-run the best seed and keep it
            out, meta = {}, {}
            del stage_rois[:], gaps[:], iou_margins[:]
            rng = np.random.default_rng(seed)
            img = _det_image(rng, 2, H, W)
            out["img"] = img
            props = []
            for i in range(2):
                pr = _jittered_proposals(rng, gts[i], 4, 5, W, H)
                pr = np.concatenate([pr, gts[i]])           # the RPN appends the gts in training (rpn/inference.py)
                props.append(pr)
                out["gt_%d" % i], out["gt_labels_%d" % i], out["props_%d" % i] = gts[i], gt_labels[i], pr
            proposals = []
            for pr in props:
                b = BoxList(torch.from_numpy(pr.copy()), (W, H))
                b.add_field("objectness", torch.linspace(0.95, 0.05, len(b)))
                proposals.append(b)
            model.train()
            for q in model.parameters():
                q.grad = None
            feats = model.Conv_Body_FPN(model.Conv_Body(torch.from_numpy(img)))
            x, result, losses = head(feats, proposals, targets())
            meta["seed"] = seed
            meta["min_iou_margin"] = min(iou_margins)
            meta["min_argmax_gap"] = min(g_ for g_ in gaps if g_ >= 0)
            meta["max_heat_prob"] = max(-g_ for g_ in gaps if g_ < 0)
            counts = [[len(b) for b in per] for per in stage_rois]
            print("seed %d: stage RoI counts %s, IoU margin %.2e, argmax gap %.2e, max heat prob %.4f"
                  % (seed, counts, meta["min_iou_margin"], meta["min_argmax_gap"], meta["max_heat_prob"]))
            refined_in_last = sum(counts[2]) - sum(len(g_) for g_ in gts)
            if trial == len(seeds):
                break
            score = min(meta["min_iou_margin"] / 3e-3, meta["min_argmax_gap"] / 3e-4) if refined_in_last >= 1 else 0.0
            if score > best[0]:
                best = (score, seed)
        assert best[0] > 0.5, "no seed gave a usable fixture"
        for k, v in losses.items():
            out["loss::" + k] = np.asarray(float(v.detach()), np.float64)
        for s, per_img in enumerate(stage_rois):
            for i, b in enumerate(per_img):
                out["stage%d_rois_%d" % (s, i)] = b
        for i, b in enumerate(result):
            out["rescore_rois_%d" % i] = b.bbox.detach().numpy()
            out["rescore_labels_%d" % i] = b.get_field("labels").numpy()
        cls_props = head.cls_loss_evaluator._proposals
        for i, b in enumerate(cls_props):
            out["cls_rois_%d" % i] = b.bbox.numpy()
            out["cls_labels_%d" % i] = b.get_field("labels").numpy()
        out["last_x"] = x.detach().numpy()[:, ::16]
        sum(losses.values()).backward()
        meta["grad_stats"] = _grad_stats(model)
    finally:
        torch.Tensor.cuda = saved_cuda
        for obj, name, fn in restore:
            setattr(obj, name, fn)
    # ---- CLSPostProcessor candidate selection (test mode), ml_nms recorded instead of run -----------------------
    rec = {}
    orig_nms = ref_inf.boxlist_ml_nms

    def recording_nms(boxlist, thresh, *a, **k):
        rec["bbox"] = boxlist.bbox.numpy().copy()
        rec["scores"] = boxlist.get_field("scores").numpy().copy()
        rec["labels"] = boxlist.get_field("labels").numpy().copy()
        rec["thresh"] = thresh
        return boxlist
    ref_inf.boxlist_ml_nms = recording_nms
    try:
        n = 60
        w_, h_ = rng.uniform(20, 500, n), rng.uniform(20, 400, n)
        x_, y_ = rng.uniform(-30, 1333 - 0.6 * w_), rng.uniform(-20, 800 - 0.6 * h_)      # some stick out: clip_to_image
        pbox = np.stack([x_, y_, x_ + w_, y_ + h_], 1).astype(np.float32)
        logits = (rng.standard_normal((n, 81)) * 2.5).astype(np.float32)
        post = ref_inf.post_processor(type="cls")
        res = post(torch.from_numpy(logits), [BoxList(torch.from_numpy(pbox.copy()), (1333, 800))])
        out["post_logits"], out["post_boxes"] = logits, pbox
        out["post_cand_bbox"], out["post_cand_scores"], out["post_cand_labels"] = rec["bbox"], rec["scores"], rec["labels"]
        prob = torch.softmax(torch.from_numpy(logits), -1).numpy()
        meta["post_score_margin"] = float(np.abs(prob - cfg.GRID_RCNN.SCORE_THRESH).min())
        meta["post_thresh"], meta["post_nms"] = float(post.score_thresh), float(rec["thresh"])
        # RSM re-scoring branch (inference.py:62-76): s^0.8 * p^0.2 on the label's probability
        bl = BoxList(torch.from_numpy(pbox.copy()), (1333, 800))
        sc = rng.uniform(0.05, 1, n).astype(np.float32)
        lb = rng.integers(1, 81, n).astype(np.int64)
        bl.add_field("scores", torch.from_numpy(sc.copy()))
        bl.add_field("labels", torch.from_numpy(lb.copy()))
        r2 = post(torch.from_numpy(logits), [bl], rescore=True)
        out["post_rs_scores_in"], out["post_rs_labels"] = sc, lb
        out["post_rs_scores_out"] = r2[0].get_field("scores").numpy()
    finally:
        ref_inf.boxlist_ml_nms = orig_nms
    np.savez_compressed(os.path.join(HERE, "model_cpm.npz"), **out)
    with open(os.path.join(HERE, "model_cpm_meta.json"), "w") as f:
        json.dump(meta, f)
    print("cpm:", {k: float(v) for k, v in out.items() if k.startswith("loss::")})
    print("stage RoI counts:", [[len(b) for b in per] for per in stage_rois], "cls", [len(b) for b in cls_props],
          "rescore", [len(b) for b in result])
    print("min IoU margin to a stage threshold: %.2e; min |score - thresh|: %.2e; candidates %d"
          % (meta["min_iou_margin"], meta["post_score_margin"], len(rec["scores"])))
    print("decode: min relative top1-top2 gap of a heat map %.2e, largest heat-map probability %.6f"
          % (meta["min_argmax_gap"], meta["max_heat_prob"]))


def gen_rpn():
    """Row a-3 as wholes: the REFERENCE RPNLossComputation.__call__ (rpn/loss.py:88-126) and
    RPNPostProcessor.forward (rpn/inference.py:67-172, training mode: per-level top-k, decode, clip, NMS, post-NMS top-n,
    the batch-wide top-k over all levels, gt append) on a 2-image batch over five FPN levels.  RPN.BATCH_SIZE_PER_IMAGE is
    raised so far that the sampler keeps every valid anchor (no random draw; the cfg key is the reference's own), and
    torchvision's nms -- absent here -- is the oracle's greedy NMS, which tests/test_oracle_golden.py pins to the
    reference's compiled soft_nms.cpp ('hard' method).  Objectness and deltas are drawn directly (continuous: no score
    ties); stored: both losses, their gradients w.r.t. every level's objectness / delta map, and the proposals."""
    import oracle.pyoracle as O
    import pet.lib.ops.nms as ref_nms_mod
    import pet.utils.data.structures.boxlist_ops as ref_bl
    from pet.rcnn.modeling.rpn.rpn import RPNModule
    from pet.utils.data.structures.bounding_box import BoxList
    from pet.utils.data.structures.image_list import ImageList

    def oracle_nms(boxes, scores, thr):
        return torch.from_numpy(O.nms(boxes.detach().numpy(), scores.detach().numpy(), float(thr))).long()
    ref_nms_mod.nms = oracle_nms
    ref_bl._box_nms = oracle_nms
    cfg = _cpm_cfg()
    cfg.RPN.BATCH_SIZE_PER_IMAGE = 1000000
    cfg.RPN.PRE_NMS_TOP_N_TRAIN = 300
    cfg.RPN.POST_NMS_TOP_N_TRAIN = 120
    cfg.RPN.FPN_POST_NMS_TOP_N_TRAIN = 260
    assert cfg.RPN.FPN_POST_NMS_PER_BATCH and cfg.RPN.MIN_SIZE == 0
    H, W, N = 160, 224, 2
    rng = np.random.default_rng(11)
    torch.manual_seed(0)
    rpn = RPNModule([256] * 5)
    rpn.train()
    A = rpn.anchor_generator.num_anchors_per_location()[0]
    shapes = [((H + s - 1) // s, (W + s - 1) // s) for s in (4, 8, 16, 32, 64)]
    feats = [torch.zeros(N, 4, h, w) for h, w in shapes]
    images = ImageList(torch.zeros(N, 3, H, W), [(H, W)] * N)
    gts = [np.array([[12, 20, 96, 130], [100, 8, 215, 90], [60, 70, 180, 150], [150, 100, 200, 155], [5, 5, 40, 44]],
                    np.float32),
           np.array([[30, 30, 190, 140], [8, 90, 70, 156], [120, 12, 160, 60]], np.float32)]
    targets = []
    for g_ in gts:
        t = BoxList(torch.from_numpy(g_.copy()), (W, H))
        t.add_field("labels", torch.ones(len(g_), dtype=torch.int64))
        targets.append(t)
    obj = [torch.from_numpy(rng.normal(0, 2.0, (N, A, h, w)).astype(np.float32)).requires_grad_(True) for h, w in shapes]
    reg = [torch.from_numpy(rng.normal(0, 0.4, (N, 4 * A, h, w)).astype(np.float32)).requires_grad_(True) for h, w in shapes]
    anchors = rpn.anchor_generator(images, feats)
    l_obj, l_box = rpn.loss_evaluator(anchors, obj, reg, targets)
    (l_obj + l_box).backward()
    with torch.no_grad():
        props = rpn.box_selector_train(anchors, obj, reg, targets)
    # how many anchors the loss looked at (the fixture is only meaningful with positives and negatives in both images)
    lab, _ = rpn.loss_evaluator.prepare_targets([ref_bl.cat_boxlist(a) for a in anchors], targets)
    out = {"H": H, "W": W, "loss_objectness": float(l_obj), "loss_rpn_box_reg": float(l_box),
           "n_pos": [int((l == 1).sum()) for l in lab], "n_neg": [int((l == 0).sum()) for l in lab],
           "pre_nms": 300, "post_nms": 120, "fpn_post_nms": 260}
    arrs = {}
    for i in range(5):
        arrs["obj%d" % i] = obj[i].detach().numpy()
        arrs["reg%d" % i] = reg[i].detach().numpy()
        arrs["dobj%d" % i] = obj[i].grad.numpy()
        arrs["dreg%d" % i] = reg[i].grad.numpy()
    for n in range(N):
        arrs["gt%d" % n] = gts[n]
        arrs["prop_box%d" % n] = props[n].bbox.numpy()
        arrs["prop_obj%d" % n] = props[n].get_field("objectness").numpy()
    assert all(p > 0 for p in out["n_pos"]) and all(q > 0 for q in out["n_neg"]), out
    # robustness margin of the fixture: the gap between the batch-wide top-k's last kept and first dropped score
    np.savez_compressed(os.path.join(HERE, "rpn_whole.npz"), **arrs)
    with open(os.path.join(HERE, "rpn_whole_meta.json"), "w") as f:
        json.dump(out, f)
    print("rpn:", out, [len(p) for p in props])



def gen_rpn_head():
    """Row a-3's HEAD in training, as a whole, against the REFERENCE: RPNHead.forward
    (rpn/rpn.py:34-41) -> RPNLossComputation.__call__ (rpn/loss.py:88-126) with the reference's own sampler at its own
    budget (RPN.BATCH_SIZE_PER_IMAGE = 256, POSITIVE_FRACTION 0.5; the draw is torch.randperm under a fixed seed and the
    sampled masks are STORED, so that the test feeds the same sample instead of a seed) -> backward: the gradients of
    the six head parameters and of the five feature maps.  This is what csrc/rpn_sparse.hip computes from the <= 512
    sampled anchors; the fixture holds it to the reference directly, not through the dense formulation."""
    import pet.utils.data.structures.boxlist_ops as ref_bl
    from pet.rcnn.modeling.rpn.rpn import RPNModule
    from pet.utils.data.structures.bounding_box import BoxList
    from pet.utils.data.structures.image_list import ImageList
    cfg = _cpm_cfg()
    assert cfg.RPN.BATCH_SIZE_PER_IMAGE == 256 and cfg.RPN.POSITIVE_FRACTION == 0.5
    H, W, N, C = 160, 224, 2, 256
    torch.manual_seed(0)
    rpn = RPNModule([C] * 5)
    rpn.train()
    det_fill_(rpn.head)
    shapes = [((H + s - 1) // s, (W + s - 1) // s) for s in (4, 8, 16, 32, 64)]
    feats = [rpn_head_feature(i, (N, C, h, w)).requires_grad_(True) for i, (h, w) in enumerate(shapes)]
    images = ImageList(torch.zeros(N, 3, H, W), [(H, W)] * N)
    gts = [np.array([[12, 20, 96, 130], [100, 8, 215, 90], [60, 70, 180, 150], [150, 100, 200, 155], [5, 5, 40, 44]],
                    np.float32),
           np.array([[30, 30, 190, 140], [8, 90, 70, 156], [120, 12, 160, 60]], np.float32)]
    targets = []
    for g_ in gts:
        t = BoxList(torch.from_numpy(g_.copy()), (W, H))
        t.add_field("labels", torch.ones(len(g_), dtype=torch.int64))
        targets.append(t)
    obj, reg = rpn.head(feats)
    anchors = rpn.anchor_generator(images, feats)
    # record the sampler's masks as the loss draws them
    drawn = {}
    sampler = rpn.loss_evaluator.fg_bg_sampler
    inner = sampler.__class__.__call__

    def recording(self, matched):
        pos, neg = inner(self, matched)
        drawn["pos"] = torch.cat(pos).numpy().astype(bool)
        drawn["neg"] = torch.cat(neg).numpy().astype(bool)
        drawn["quota"] = np.array([[int(p.sum()), int(q.sum())] for p, q in zip(pos, neg)], np.int32)
        return pos, neg
    sampler.__class__.__call__ = recording
    try:
        torch.manual_seed(7)
        l_obj, l_box = rpn.loss_evaluator(anchors, obj, reg, targets)
    finally:
        sampler.__class__.__call__ = inner
    (l_obj + l_box).backward()
    out = {"H": H, "W": W, "C": C, "loss_objectness": float(l_obj), "loss_rpn_box_reg": float(l_box),
           "quota": drawn["quota"].tolist(), "grad_stats": {}}
    assert all(q[0] > 0 and q[1] > 0 for q in out["quota"]), out["quota"]
    arrs = {"pos": drawn["pos"], "neg": drawn["neg"], "quota": drawn["quota"]}
    for n in range(N):
        arrs["gt%d" % n] = gts[n]
    for i in range(5):
        arrs["obj%d" % i] = obj[i].detach().numpy()
        arrs["reg%d" % i] = reg[i].detach().numpy()
        g_ = feats[i].grad
        out["grad_stats"]["feat%d" % i] = [float(g_.double().abs().sum()), float((g_.double() ** 2).sum())]
        arrs["dfeat%d" % i] = g_[:, ::8].numpy() if i == 0 else g_.numpy()      # (the finest map: every 8th channel)
    for k, q in rpn.head.named_parameters():
        g_ = q.grad
        out["grad_stats"][k] = [float(g_.double().abs().sum()), float((g_.double() ** 2).sum())]
        arrs["dparam::" + k] = g_[::2, ::2].numpy() if k == "conv.weight" else g_.numpy()
    np.savez_compressed(os.path.join(HERE, "rpn_head.npz"), **arrs)
    with open(os.path.join(HERE, "rpn_head_meta.json"), "w") as f:
        json.dump(out, f)
    print("rpn_head:", {k: v for k, v in out.items() if k != "grad_stats"},
          os.path.getsize(os.path.join(HERE, "rpn_head.npz")) // 1024, "KB")


def gen_cocoeval():
    """Row f-3: the reference's vendored COCOeval (pet/rcnn/datasets/mycocoeval.py:62-423, bbox
    protocol) run here on a synthetic dataset.  Its only third-party call is pycocotools' compiled box IoU
    (mycocoeval.py:190, maskUtils.iou); the harness supplies that one function in numpy (intersection over union of
    [x, y, w, h] boxes, over the detection's area against a crowd) and a minimal in-memory COCO index with the four
    accessors COCOeval uses (getImgIds / getCatIds / getAnnIds / loadAnns).  Stored: the dataset, the detections, the 16
    summary numbers and the precision / recall arrays."""
    import importlib

    def iou(d, g, iscrowd):
        d, g = np.asarray(d, np.float64).reshape(-1, 4), np.asarray(g, np.float64).reshape(-1, 4)
        out = np.zeros((len(d), len(g)))
        for j in range(len(g)):
            for i in range(len(d)):
                w = min(d[i, 0] + d[i, 2], g[j, 0] + g[j, 2]) - max(d[i, 0], g[j, 0])
                h = min(d[i, 1] + d[i, 3], g[j, 1] + g[j, 3]) - max(d[i, 1], g[j, 1])
                if w <= 0 or h <= 0:
                    continue
                inter = w * h
                da, ga = d[i, 2] * d[i, 3], g[j, 2] * g[j, 3]
                out[i, j] = inter / (da if iscrowd[j] else da + ga - inter)
        return out
    sys.modules["pycocotools.mask"].iou = iou
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    # the file on its own, where it lies (pet/rcnn/datasets/__init__.py pulls in the torchvision-based loaders)
    ref = _load_ref_file("pet/rcnn/datasets/mycocoeval.py", "ref_mycocoeval")
    # numpy 2 refuses the float sample count the file passes to np.linspace (np.round(..) + 1, mycocoeval.py Params):
    # the numpy of the reference's day truncated it -- same stand-in class as np.float = float above
    _linspace = np.linspace
    np.linspace = lambda start, stop, num=50, **kw: _linspace(start, stop, int(num), **kw)

    class MiniCOCO(object):
        def __init__(self, images, cats, anns):
            self.images, self.cats, self.anns = images, cats, {a["id"]: a for a in anns}

        def getImgIds(self):
            return [im["id"] for im in self.images]

        def getCatIds(self):
            return [c["id"] for c in self.cats]

        def getAnnIds(self, imgIds=[], catIds=[]):
            return [a["id"] for a in self.anns.values()
                    if (not len(imgIds) or a["image_id"] in imgIds) and (not len(catIds) or a["category_id"] in catIds)]

        def loadAnns(self, ids):
            return [self.anns[i] for i in ids]

    rng = np.random.default_rng(5)
    images = [{"id": i, "width": 640, "height": 480} for i in range(1, 15)]
    cats = [{"id": c, "name": "c%d" % c} for c in (1, 2, 5)]
    anns, dets = [], []
    for im in images:
        for c in cats:
            n_gt = int(rng.integers(0, 5))
            for _ in range(n_gt):
                side = float(rng.choice([12.0, 40.0, 150.0])) * float(rng.uniform(0.7, 1.4))
                w, h = side * float(rng.uniform(0.6, 1.6)), side
                x, y = float(rng.uniform(0, 640 - w)), float(rng.uniform(0, 480 - h))
                crowd = int(rng.uniform() < 0.12)
                anns.append({"id": len(anns) + 1, "image_id": im["id"], "category_id": c["id"], "bbox": [x, y, w, h],
                             "area": w * h * float(rng.uniform(0.5, 1.0)), "iscrowd": crowd})
                for _ in range(int(rng.integers(0, 3))):          # jittered copies: true positives at various IoU
                    j = rng.normal(0, 0.12, 4)
                    dets.append({"image_id": im["id"], "category_id": c["id"],
                                 "bbox": [x + j[0] * w, y + j[1] * h, w * float(np.exp(j[2])), h * float(np.exp(j[3]))],
                                 "score": float(rng.uniform(0.05, 1.0))})
            for _ in range(int(rng.integers(0, 4))):              # false positives
                w, h = float(rng.uniform(8, 300)), float(rng.uniform(8, 300))
                dets.append({"image_id": im["id"], "category_id": c["id"],
                             "bbox": [float(rng.uniform(0, 640 - w)), float(rng.uniform(0, 480 - h)), w, h],
                             "score": float(rng.uniform(0.0, 0.8))})
    # one (image, category) cell beyond maxDets = 100
    for _ in range(130):
        w, h = float(rng.uniform(8, 120)), float(rng.uniform(8, 120))
        dets.append({"image_id": 3, "category_id": 2, "bbox": [float(rng.uniform(0, 500)), float(rng.uniform(0, 350)), w, h],
                     "score": float(rng.uniform(0.0, 1.0))})
    gt_json = {"images": images, "categories": cats, "annotations": anns}
    dt_anns = []
    for i, d in enumerate(dets):                                  # what COCO.loadRes adds to bbox results
        a = dict(d)
        a["id"], a["area"], a["iscrowd"] = i + 1, d["bbox"][2] * d["bbox"][3], 0
        dt_anns.append(a)
    import copy
    E = ref.COCOeval(MiniCOCO(images, cats, copy.deepcopy(anns)), MiniCOCO(images, cats, dt_anns), "bbox")
    E.evaluate()
    E.accumulate()
    E.summarize()
    np.linspace = _linspace
    with open(os.path.join(HERE, "cocoeval_ref.json"), "w") as f:
        json.dump({"gt": gt_json, "dt": dets, "stats": [float(v) for v in E.stats]}, f)
    np.savez_compressed(os.path.join(HERE, "cocoeval_ref.npz"), precision=E.eval["precision"], recall=E.eval["recall"])
    print("cocoeval:", len(anns), "gts", len(dets), "dets", [round(float(v), 4) for v in E.stats])



def gen_soft_nms(ref_ext):
    """soft_nms_cpu of the reference (csrc/NMS/soft_nms.cpp compiled into oracle/_ref) on seeded box sets: all three
    methods, ties in the scores, heavy overlap (many removals), n = 0 / 1."""
    out = {}
    rng = np.random.default_rng(42)
    cases = []
    for n, span, method, sigma, thr, ms in [(200, 300, 1, 0.5, 0.3, 0.001), (200, 300, 2, 0.5, 0.3, 0.001),
                                            (200, 300, 0, 0.5, 0.5, 0.001), (500, 150, 1, 0.5, 0.3, 0.05),
                                            (500, 150, 2, 0.3, 0.3, 0.05), (64, 60, 1, 0.5, 0.1, 0.2),
                                            (1, 50, 1, 0.5, 0.3, 0.001), (0, 50, 1, 0.5, 0.3, 0.001),
                                            (1500, 400, 1, 0.5, 0.3, 0.0001),
                                            # hard method = the reference's own CPU greedy NMS: pins orc_nms / the HIP NMS
                                            (1000, 300, 0, 0.5, 0.7, 0.001), (2000, 500, 0, 0.5, 0.3, 0.001),
                                            (700, 120, 0, 0.5, 0.5, 0.001)]:
        xy = rng.uniform(0, span, (n, 2))
        wh = rng.uniform(4, 120, (n, 2))
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        scores = rng.uniform(0, 1, n).astype(np.float32)
        if n > 10 and not (method == 0 and n >= 700):
            scores[5:9] = scores[4]                      # exact ties: the first position wins
            boxes[10] = boxes[3]                         # duplicate box
        cases.append((boxes, scores, method, sigma, thr, ms))
    for i, (boxes, scores, method, sigma, thr, ms) in enumerate(cases):
        d, s_, k = ref_ext.soft_nms_cpu(torch.from_numpy(boxes.copy()), torch.from_numpy(scores.copy()), thr, method,
                                        sigma, ms)
        out["c%d_boxes" % i], out["c%d_scores" % i] = boxes, scores
        out["c%d_cfg" % i] = np.array([method, sigma, thr, ms], np.float64)
        out["c%d_out_boxes" % i], out["c%d_out_scores" % i], out["c%d_out_idx" % i] = d.numpy(), s_.numpy(), k.numpy()
    # multi-label twin (NMS/ml_soft_nms.cpp): labels 1..5, with and without topk; the hard method is the reference's
    # own CPU form of ml_nms
    ml = []
    for n, span, method, sigma, thr, ms, topk in [(400, 200, 0, 0.5, 0.5, 0.001, -1), (900, 300, 0, 0.5, 0.3, 0.001, -1),
                                                  (400, 200, 1, 0.5, 0.3, 0.01, -1), (400, 200, 2, 0.5, 0.3, 0.01, -1),
                                                  (400, 200, 1, 0.5, 0.3, 0.01, 50), (400, 200, 0, 0.5, 0.5, 0.001, 30),
                                                  (50, 100, 1, 0.5, 0.3, 0.01, 0), (0, 100, 1, 0.5, 0.3, 0.01, -1)]:
        xy = rng.uniform(0, span, (n, 2))
        wh = rng.uniform(4, 120, (n, 2))
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        scores = rng.uniform(0, 1, n).astype(np.float32)
        labels = rng.integers(1, 6, n).astype(np.int64)
        ml.append((boxes, scores, labels, method, sigma, thr, ms, topk))
    for i, (boxes, scores, labels, method, sigma, thr, ms, topk) in enumerate(ml):
        d, s_, l_, k = ref_ext.ml_soft_nms_cpu(torch.from_numpy(boxes.copy()), torch.from_numpy(scores.copy()),
                                               torch.from_numpy(labels.copy()), thr, method, sigma, ms, topk)
        out["m%d_boxes" % i], out["m%d_scores" % i], out["m%d_labels" % i] = boxes, scores, labels
        out["m%d_cfg" % i] = np.array([method, sigma, thr, ms, topk], np.float64)
        out["m%d_out_boxes" % i], out["m%d_out_scores" % i] = d.numpy(), s_.numpy()
        out["m%d_out_labels" % i], out["m%d_out_idx" % i] = l_.numpy(), k.numpy()
    print("ml_soft_nms kept", [int(out["m%d_out_idx" % i].shape[0]) for i in range(len(ml))])
    np.savez_compressed(os.path.join(HERE, "soft_nms.npz"), **out)
    print("soft_nms:", len(cases), "cases; kept", [int(out["c%d_out_idx" % i].shape[0]) for i in range(len(cases))])


def _load_ref_file(rel, name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_data_pipeline():
    """Index sequences of the reference's samplers and output sizes of its Resize for seeded cases (host logic of the
    input pipeline).  torchvision.transforms is replaced by an inert stand-in: only Resize.get_size
    (pure Python) is called."""
    import random
    tvt = types.ModuleType("torchvision.transforms")
    tvt.ColorJitter = object
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvt.functional = tvf
    sys.modules["torchvision"].transforms = tvt
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.transforms.functional"] = tvf
    T = _load_ref_file("pet/utils/data/transforms/transforms.py", "ref_transforms")
    DS = _load_ref_file("pet/utils/data/samplers/distributed.py", "ref_distributed").DistributedSampler
    GB = _load_ref_file("pet/utils/data/samplers/grouped_batch_sampler.py", "ref_grouped").GroupedBatchSampler
    IB = _load_ref_file("pet/utils/data/samplers/iteration_based_batch_sampler.py", "ref_iter").IterationBasedBatchSampler
    out = {"get_size": [], "distributed": [], "grouped": [], "iteration": []}
    rng = random.Random(11)
    for i in range(60):
        w, h = rng.randint(60, 2000), rng.randint(60, 2000)
        mins, mx = [(800,), (600,), (640, 672, 704, 736, 768, 800), (1200,)][i % 4], [1333, 1000, 1333, 2000][i % 4]
        random.seed(100 + i)
        out["get_size"].append([w, h, list(mins), mx, 100 + i, list(T.Resize(mins, mx).get_size((w, h)))])
    out["get_size"].append([1333, 800, [800], 1333, 1, list(T.Resize((800,), 1333).get_size((1333, 800)))])
    out["get_size"].append([640, 480, [800], 1333, 1, list(T.Resize((800,), 1333).get_size((640, 480)))])
    for n, world, epoch, shuffle in [(23, 4, 0, True), (23, 4, 5, True), (16, 8, 2, True), (10, 3, 0, False)]:
        for r in range(world):
            smp = DS(list(range(n)), num_replicas=world, rank=r, shuffle=shuffle)
            smp.set_epoch(epoch)
            out["distributed"].append([n, world, r, epoch, shuffle, list(smp)])
    for n, world, r, epoch, bs, drop, seed in [(37, 2, 1, 3, 2, False, 0), (37, 2, 0, 3, 3, True, 1), (19, 2, 1, 1, 2, False, 2),
                                               (64, 1, 0, 7, 4, False, 3)]:
        g = torch.Generator().manual_seed(seed)
        gids = torch.randint(0, 2, (n,), generator=g).tolist()
        smp = DS(list(range(n)), num_replicas=world, rank=r, shuffle=True)
        smp.set_epoch(epoch)
        b = GB(smp, gids, bs, drop_uneven=drop)
        out["grouped"].append([n, world, r, epoch, bs, drop, gids, [list(map(int, x)) for x in b], len(b)])
    for n, bs, iters, start, seed in [(13, 2, 25, 7, 4), (13, 3, 9, 0, 5)]:
        g = torch.Generator().manual_seed(seed)
        gids = torch.randint(0, 2, (n,), generator=g).tolist()
        smp = DS(list(range(n)), num_replicas=1, rank=0, shuffle=True)
        it = IB(GB(smp, gids, bs), iters, start)
        out["iteration"].append([n, bs, iters, start, gids, [list(map(int, x)) for x in it]])
    # suffix matching of checkpoint keys (pet/utils/checkpointer.py:190-242): every tensor carries its own index
    from pet.utils.checkpointer import align_and_update_state_dicts, strip_prefix_if_present
    model_keys = ["Conv_Body.conv1.weight", "Conv_Body.bn1.weight", "Conv_Body.layer1.0.conv1.weight",
                  "Conv_Body.layer1.0.bn1.running_mean", "Conv_Body.layer1.0.downsample.0.weight",
                  "Conv_Body_FPN.p5_in.weight", "RPN.head.conv.weight", "Grid_Cascade_RCNN.Head_cls.fc6.weight",
                  "Conv_Body.layer2.3.conv3.weight", "Conv_Body.layer1.0.conv1.bias"]
    weight_keys = ["conv1.weight", "bn1.weight", "layer1.0.conv1.weight", "layer1.0.bn1.running_mean",
                   "layer1.0.downsample.0.weight", "layer2.3.conv3.weight", "fc.weight", "0.conv1.weight", "weight",
                   "fc6.weight", "head.conv.weight"]
    msd = {k: torch.tensor(-1.0) for k in model_keys}
    wd = {k: torch.tensor(float(i)) for i, k in enumerate(weight_keys)}
    upd, mismatch = align_and_update_state_dicts(msd, wd, -1)
    out["align"] = {"model_keys": model_keys, "weight_keys": weight_keys,
                    "picked": {k: int(v) for k, v in upd.items()}, "mismatch": sorted(mismatch)}
    pref = {"module.a.w": 1, "module.b": 2}
    out["strip"] = [[list(pref), list(strip_prefix_if_present(dict(pref), "module."))],
                    [["module.a", "b"], list(strip_prefix_if_present({"module.a": 1, "b": 2}, "module."))]]
    with open(os.path.join(HERE, "data_pipeline.json"), "w") as f:
        json.dump(out, f)
    print("data pipeline:", {k: len(v) for k, v in out.items()})


def main():
    ref_ext = build_ref()
    assert ref_ext is not None, "needs /root/reference"
    if sys.argv[1:] == ["softnms"]:
        return gen_soft_nms(ref_ext)
    install_standins(ref_ext)
    if sys.argv[1:] == ["x101"]:
        return gen_x101_meta()
    if sys.argv[1:] == ["r101"]:
        return gen_r101_meta()
    if sys.argv[1:] == ["data"]:
        return gen_data_pipeline()
    if sys.argv[1:] == ["cascade"]:
        return gen_cascade()
    if sys.argv[1:] == ["big"]:
        return gen_model_big()
    if sys.argv[1:] == ["cpm"]:
        return gen_cpm_train()
    if sys.argv[1:] == ["rpn"]:
        return gen_rpn()
    if sys.argv[1:] == ["rpn_head"]:
        return gen_rpn_head()
    if sys.argv[1:] == ["cocoeval"]:
        return gen_cocoeval()
    cfg = load_cfg("cfgs/rcnn/mscoco/grid_cascade/iou_helper/rescore/e2e_grid_cascade@567_rcnn_R-50-FPN_2x.yaml")
    cfg.DEVICE = "cpu"
    ops = {}
    gen_ops(ref_ext, ops)
    gen_grid(ops)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops)
    model, meta = {}, {}
    gen_model(model, meta)
    np.savez_compressed(os.path.join(HERE, "model_r50.npz"), **model)
    with open(os.path.join(HERE, "model_r50_meta.json"), "w") as f:
        json.dump(meta, f)
    print("ops:", len(ops), "arrays; model:", len(model), "arrays")


if __name__ == "__main__":
    main()
