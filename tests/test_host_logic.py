"""CPU-only: the device-agnostic host logic of the pet.* mirror against vectors captured from the reference
(tests/golden/ops.npz) and against the oracle.  No HIP kernel is launched here."""
import json
import os

import numpy as np
import pytest
import torch

R50_YAML = ("/root/reference/cfgs/rcnn/mscoco/grid_cascade/iou_helper/rescore/"
            "e2e_grid_cascade@567_rcnn_R-50-FPN_2x.yaml")

CPM_OPTS = ["MODEL.FPN_ON", True, "MODEL.FASTER_RCNN", False, "MODEL.GRID_ON", True, "MODEL.NUM_CLASSES", 81,
            "MODEL.CONV1_RGB2BGR", False, "RPN.ANCHOR_STRIDE", (4, 8, 16, 32, 64), "RPN.PRE_NMS_TOP_N_TRAIN", 2000,
            "RPN.PRE_NMS_TOP_N_TEST", 1000, "RPN.POST_NMS_TOP_N_TEST", 1000, "RPN.FPN_POST_NMS_TOP_N_TEST", 1000,
            "GRID_RCNN.NMS", 0.3, "GRID_RCNN.SCORE_THRESH", 0.03, "GRID_RCNN.FUSED_ON", False,
            "GRID_RCNN.IOU_HELPER", True, "GRID_RCNN.IOU_HELPER_MERGE", True, "GRID_RCNN.RESCORE_ON", True,
            "GRID_RCNN.CASCADE_MAPPING_ON", True, "GRID_RCNN.CASCADE_MAPPING_OPTION.TEST_ENSEMBLE", False]


@pytest.fixture()
def cpm_cfg():
    from pet.rcnn.core import config
    config.reset_cfg()
    config.merge_cfg_from_list(CPM_OPTS)
    yield config.cfg
    config.reset_cfg()


def test_cfg_yaml_merge(tmp_path):
    from pet.rcnn.core import config
    config.reset_cfg()
    y = tmp_path / "c.yaml"
    y.write_text("MODEL:\n  FPN_ON: True\n  NUM_CLASSES: 81\nBACKBONE:\n  RESNET:\n    LAYERS: (3, 4, 23, 3)\n"
                 "RPN:\n  ANCHOR_STRIDE: (4, 8, 16, 32, 64)\nSOLVER:\n  STEPS: [120000, 160000]\n  BASE_LR: 0.02\n"
                 "GRID_RCNN:\n  CASCADE_MAPPING_OPTION:\n    FG_IOU_THRESHOLD: (0.5, 0.6, 0.7)\n"
                 "PIXEL_MEANS: [102.9801, 115.9465, 122.7717]\n")
    config.merge_cfg_from_file(str(y))
    c = config.cfg
    assert c.MODEL.FPN_ON is True and c.BACKBONE.RESNET.LAYERS == (3, 4, 23, 3)
    assert c.GRID_RCNN.CASCADE_MAPPING_OPTION.FG_IOU_THRESHOLD == [0.5, 0.6, 0.7]      # tuple -> list coercion
    assert isinstance(c.PIXEL_MEANS, np.ndarray) and c.SOLVER.STEPS == [120000, 160000]
    config.merge_cfg_from_list(["SOLVER.BASE_LR", "0.01", "TRAIN.SCALES", "(800,)"])
    assert c.SOLVER.BASE_LR == 0.01 and c.TRAIN.SCALES == (800,)
    with pytest.raises(KeyError):
        bad = tmp_path / "bad.yaml"
        bad.write_text("NO_SUCH_KEY: 1\n")
        config.merge_cfg_from_file(str(bad))
    with pytest.raises(ValueError):
        config.merge_cfg_from_list(["SOLVER.BASE_LR", "abc"])
    config.reset_cfg()


@pytest.mark.skipif(not os.path.exists(R50_YAML), reason="reference tree not present on this box")
def test_every_reference_yaml_merges():
    """All 55 files under cfgs/rcnn/mscoco merge into the config tree unchanged ("cfgs/rcnn/*.yaml run
    unchanged"), including the nine that set VIS keys (config.py:1143-1276)."""
    import glob
    from pet.rcnn.core import config
    files = sorted(glob.glob("/root/reference/cfgs/rcnn/mscoco/**/*.yaml", recursive=True))
    assert len(files) >= 55
    try:
        for f in files:
            config.reset_cfg()
            config.merge_cfg_from_file(f)
            assert config.cfg.MODEL.NUM_CLASSES == 81, f
    finally:
        config.reset_cfg()


def test_every_reference_ops_name_imports():
    """pet/lib/ops/__init__.py:1-30 of the reference: every exported name exists here (callers type-check against
    them); the ones outside the hot path raise on use, never silently fall back."""
    import pet.lib.ops as ops
    names = ["nms", "ml_nms", "nms_rotated", "poly_nms", "soft_nms", "ml_soft_nms", "box_voting", "box_ml_voting",
             "box_iou", "box_iou_rotated", "l2_loss", "IOULoss", "BoundedIoULoss", "MaskIOULoss", "DICELoss",
             "smooth_l1_loss", "smooth_l1_loss_LW", "SigmoidFocalLoss", "equalization_loss", "LovaszHinge",
             "LovaszSoftmax", "lovasz_softmax_loss", "LabelSmoothing", "FrozenBatchNorm2d", "NaiveSyncBatchNorm",
             "Conv2dSamePadding", "Conv2dWS", "SplAtConv2d", "DeformConv", "ModulatedDeformConv", "DeformConvPack",
             "ModulatedDeformConvPack", "L2Norm", "MixtureBatchNorm2d", "MixtureGroupNorm", "Mish", "H_Swish",
             "H_Sigmoid", "Swish", "SwishX", "DropBlock2D", "Scale", "SeConv2d", "GlobalContextBlock", "ECA",
             "PoolPointsInterp", "roi_align", "ROIAlign", "roi_align_rotated", "ROIAlignRotated", "roi_pool", "ROIPool",
             "AffineChannel2d"]
    for n in names:
        assert hasattr(ops, n), n
    assert isinstance(ops.MixtureBatchNorm2d, type) and isinstance(ops.ModulatedDeformConvPack, type)
    with pytest.raises(RuntimeError, match="outside the CPM R-CNN hot path"):
        ops.MixtureBatchNorm2d(8)
    with pytest.raises(RuntimeError, match="outside the CPM R-CNN hot path"):
        ops.roi_pool(None, None, (7, 7), 1.0)
    import torch
    bn = ops.FrozenBatchNorm2d(4)
    assert sorted(bn.state_dict()) == ["bias", "running_mean", "running_var", "weight"]
    assert torch.allclose(bn(torch.ones(1, 4, 2, 2)), torch.ones(1, 4, 2, 2), atol=1e-4)


@pytest.mark.skipif(not os.path.exists(R50_YAML), reason="reference tree not present on this box")
def test_reference_yamls_parse_unchanged():
    from pet.rcnn.core import config
    base = os.path.dirname(R50_YAML)
    for f in [R50_YAML, base + "/backbone/e2e_grid_cascade@567_rcnn_R-101-FPN_2x.yaml",
              base + "/backbone/e2e_grid_cascade@567_rcnn_X-101b-64x4d-FPN-DCN_2x.yaml"]:
        config.reset_cfg()
        config.merge_cfg_from_file(f)
        assert config.cfg.GRID_RCNN.CASCADE_MAPPING_ON and config.cfg.GRID_RCNN.RESCORE_ON
    config.reset_cfg()
    config.merge_cfg_from_file(R50_YAML)
    ref = dict(zip(CPM_OPTS[0::2], CPM_OPTS[1::2]))
    for k, v in ref.items():
        node = config.cfg
        for part in k.split("."):
            node = node[part]
        assert node == v, k
    config.reset_cfg()
    # the offset-regression cascade family (plain / ISM / RSM / ISM+RSM)
    cas = "/root/reference/cfgs/rcnn/mscoco/cascade/"
    for sub in ("", "ISM/", "RSM/", "ISM+RSM/"):
        config.reset_cfg()
        config.merge_cfg_from_file(cas + sub + "e2e_cascade_rcnn@2_R-50-FPN_1x.yaml")
        assert config.cfg.MODEL.CASCADE_ON and config.cfg.MODEL.FASTER_RCNN and config.cfg.CASCADE_RCNN.NUM_STAGE == 2
    ref = dict(zip(CASCADE_OPTS[0::2], CASCADE_OPTS[1::2]))
    for k, v in ref.items():
        node = config.cfg
        for part in k.split("."):
            node = node[part]
        assert node == v, k
    config.reset_cfg()


def test_anchor_generator(golden_ops, cpm_cfg):
    from pet.rcnn.modeling.rpn.anchor_generator import AnchorGenerator, generate_anchors, make_anchor_generator
    g = golden_ops
    assert np.array_equal(generate_anchors(16, (128, 256, 512), (0.5, 1, 2)).float().numpy(), g["anchors_matlab_table"])
    ag = make_anchor_generator()
    assert isinstance(ag, AnchorGenerator) and ag.num_anchors_per_location() == [3] * 5
    for i, c in enumerate(ag.cell_anchors):
        assert np.array_equal(c.numpy(), g["cell_anchors_%d" % i])
    grids = ag.grid_anchors([(5, 7), (3, 4), (2, 2), (1, 2), (1, 1)])
    for i, a in enumerate(grids):
        assert np.array_equal(a.numpy(), g["grid_anchors_%d" % i])
    assert np.array_equal(ag.visibility(grids[0], 28, 20).numpy(), g["grid_anchors_0_visibility"])
    assert list(ag.state_dict().keys()) == ["cell_anchors.%d" % i for i in range(5)]


def test_box_coder_matcher_iou_levelmapper(golden_ops):
    from pet.rcnn.utils.box_coder import BoxCoder
    from pet.rcnn.utils.matcher import Matcher
    from pet.rcnn.utils.poolers import LevelMapper
    from pet.utils.data.structures.bounding_box import BoxList
    from pet.utils.data.structures.boxlist_ops import boxlist_iou
    g = golden_ops
    t = torch.from_numpy
    bc = BoxCoder((1., 1., 1., 1.))
    assert np.array_equal(bc.decode(t(g["bc_codes"]), t(g["bc_boxes"])).numpy(), g["bc_decode"])
    assert np.array_equal(bc.encode(t(g["bc_gt"]), t(g["bc_boxes"])).numpy(), g["bc_encode"])
    iou = boxlist_iou(BoxList(t(g["iou_gt"]), (1333, 800)), BoxList(t(g["iou_props"]), (1333, 800)))
    assert np.array_equal(iou.numpy(), g["iou_out"])
    assert np.array_equal(Matcher(0.7, 0.3, True)(iou.clone()).numpy(), g["match_rpn"])
    assert np.array_equal(Matcher(0.5, 0.5, False)(iou.clone()).numpy(), g["match_cls"])
    assert np.array_equal(Matcher(0.7, 0.7, False)(iou.clone()).numpy(), g["match_g2"])
    lv = LevelMapper(2, 5)([BoxList(t(g["lvl_boxes"]), (1333, 800))])
    assert np.array_equal(lv.numpy(), g["lvl_out"])
    with pytest.raises(ValueError):
        Matcher(0.5, 0.5)(torch.zeros(0, 4))


def test_losses(golden_ops):
    from pet.lib.ops.losses import l2_loss, smooth_l1_loss
    g = golden_ops
    t = torch.from_numpy
    assert np.array_equal(smooth_l1_loss(t(g["sl1_a"]), t(g["sl1_b"]), beta=1. / 9, reduction="sum").numpy(), g["sl1_out"])
    assert np.array_equal(l2_loss(t(g["l2_x"]), t(g["l2_t"])).numpy(), g["l2_out"])


@pytest.mark.parametrize("stage,ratio", [(0, 1.0), (1, 0.5), (2, 0.25)])
def test_grid_targets_and_decoder(golden_ops, cpm_cfg, stage, ratio):
    from pet.rcnn.modeling.grid_cascade_rcnn.inference import post_processor
    from pet.rcnn.modeling.grid_cascade_rcnn.loss import loss_evaluator
    from pet.rcnn.modeling.grid_rcnn.loss import calc_sub_regions
    from pet.utils.data.structures.bounding_box import BoxList
    g = golden_ops
    assert np.array_equal(np.array(calc_sub_regions(9, 3, 56), np.int32), g["sub_regions"])
    ev = loss_evaluator(stage=stage, type="grid")
    ev.pos_result = (torch.from_numpy(g["grid_boxes"]), torch.from_numpy(g["grid_gt"]))
    tg = ev.prepare_target()
    assert np.array_equal(tg.numpy(), g["grid_targets_s%d" % stage])          # bit-exact rasterisation
    pp = post_processor(stage=stage, type="grid")
    boxes = pp.get_boxes(BoxList(torch.from_numpy(g["grid_boxes"].copy()), (1333, 800)),
                         torch.from_numpy(g["grid_logits_s%d" % stage]), False)
    np.testing.assert_allclose(boxes.numpy(), g["grid_decode_s%d" % stage], rtol=1e-6, atol=1e-4)


def test_filter_boxes_and_sampler(cpm_cfg):
    from pet.rcnn.modeling.grid_cascade_rcnn.inference import post_processor
    from pet.rcnn.utils.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    from pet.utils.data.structures.bounding_box import BoxList
    pp = post_processor(stage=0, type="grid")
    gt = BoxList(torch.tensor([[10., 20., 50., 60.], [0., 0., 5., 5.]]), (100, 100))
    pr = BoxList(torch.tensor([[10., 20., 50., 60.], [10., 21., 50., 60.], [0., 0., 5., 5.], [1., 2., 3., 4.]]), (100, 100))
    assert pp._filter_boxes(pr, gt).tolist() == [1, 3]
    torch.manual_seed(0)
    m = torch.tensor([1] * 10 + [0] * 500 + [-1] * 20)
    pos, neg = BalancedPositiveNegativeSampler(64, 0.25)([m])
    assert pos[0].dtype == torch.bool and int(pos[0].sum()) == 10 and int(neg[0].sum()) == 54
    assert not bool((pos[0] & neg[0]).any()) and not bool(pos[0][510:].any() | neg[0][510:].any())


def test_model_state_dict_abi(cpm_cfg):
    """Same module tree as the reference: every state-dict key and shape, and the same trainable set
    (tests/golden/model_r50_meta.json was dumped from the reference model)."""
    import json
    from conftest import ROOT
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "model_r50_meta.json")))
    model = convert_bn2affine_model(Generalized_RCNN(is_train=True))
    got = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    assert got == meta["state_dict"]
    assert [k for k, p in model.named_parameters() if p.requires_grad] == meta["trainable"]
    n = sum(p.numel() for p in model.parameters())
    assert abs(n - 153.89e6) < 0.02e6


X101_OPTS = ["BACKBONE.CONV_BODY", "resnext", "BACKBONE.RESNEXT.LAYERS", (3, 4, 23, 3),
             "BACKBONE.RESNEXT.STAGE_WITH_CONV", ("normal", "deform", "deform", "deform"), "BACKBONE.RESNEXT.C", 64,
             "BACKBONE.RESNEXT.WIDTH", 4, "GRID_RCNN.MAX_SAMPLE_NUM_GRID", 32]


def test_x101_dcn_state_dict_abi(cpm_cfg):
    """BASELINE config #5 (X-101-64x4d-FPN + DCN): the module tree, every key / shape (incl. the
    `conv2.conv_offset.{weight,bias}` children of DeformConvPack) and the trainable set equal the reference's
    (tests/golden/model_x101_meta.json, dumped from the reference model by make_golden.py x101)."""
    import json
    from conftest import ROOT
    from pet.rcnn.core import config
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    import pet.lib.ops as ops
    config.merge_cfg_from_list(X101_OPTS)
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "model_x101_meta.json")))
    model = convert_bn2affine_model(Generalized_RCNN(is_train=True))
    got = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    assert got == meta["state_dict"]
    assert [k for k, p in model.named_parameters() if p.requires_grad] == meta["trainable"]
    packs = [m for m in model.modules() if isinstance(m, ops.DeformConvPack)]
    assert len(packs) == 4 + 23 + 3
    assert all(float(m.conv_offset.weight.detach().abs().max()) == 0 and float(m.conv_offset.bias.detach().abs().max()) == 0
               for m in packs)                                    # resnext.py:248-252


def test_r101_state_dict_abi(cpm_cfg):
    """BASELINE config #4 (R-101-FPN, LAYERS (3, 4, 23, 3)): every key / shape and the trainable set equal the
    reference's (tests/golden/model_r101_meta.json, dumped from the reference model by make_golden.py r101)."""
    import json
    from conftest import ROOT
    from pet.rcnn.core import config
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    config.merge_cfg_from_list(["BACKBONE.RESNET.LAYERS", (3, 4, 23, 3)])
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "model_r101_meta.json")))
    model = convert_bn2affine_model(Generalized_RCNN(is_train=True))
    assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == meta["state_dict"]
    assert [k for k, p in model.named_parameters() if p.requires_grad] == meta["trainable"]
    assert len(meta["state_dict"]) == 471 and len(meta["trainable"]) == 247


def test_balanced_sample_quotas_oracle():
    """oracle/pyoracle.py balanced_sample_quotas against hand-worked cases of
    pet/rcnn/utils/balanced_positive_negative_sampler.py:36-46 (the GPU test checks cpm_sample_pos_neg against it)."""
    import oracle.pyoracle as po
    lab = np.array([1, 2, 0, 0, -1, 0,   0, 0, 0, -1,   3, 3, 3, 3, 0], dtype=np.int64)
    q = po.balanced_sample_quotas(lab, [6, 4, 5], 4, 0.5)
    assert q.tolist() == [[2, 2], [0, 3], [2, 1]]
    q = po.balanced_sample_quotas(lab.astype(np.float32), [6, 4, 5], 512, 0.25)
    assert q.tolist() == [[2, 3], [0, 3], [4, 1]]
    assert po.balanced_sample_quotas(lab[:0], [0], 4, 0.5).tolist() == [[0, 0]]


def test_l2_loss_nosync_matches_reference_quirk(golden_ops):
    """The sync-free restatement of l2_loss (used by the fused cascade path) against the bit-exact one, including
    the reference's row/column index quirk, on the golden input and on random targets with zeros."""
    from pet.lib.ops.losses import l2_loss, l2_loss_nosync
    g = golden_ops
    x, t = torch.from_numpy(g["l2_x"]), torch.from_numpy(g["l2_t"])
    assert abs(float(l2_loss_nosync(x, t)) - float(l2_loss(x, t))) <= 1e-6 * abs(float(l2_loss(x, t)))
    gen = torch.Generator().manual_seed(3)
    for _ in range(5):
        x = torch.randn(37, 2, generator=gen)
        fg = torch.rand(37, generator=gen)
        fg[torch.rand(37, generator=gen) < 0.3] = 0.0
        fg[torch.rand(37, generator=gen) < 0.1] = 1.0
        t = torch.stack([1 - fg, fg], 1)
        a, b = float(l2_loss_nosync(x, t)), float(l2_loss(x, t))
        assert abs(a - b) <= 1e-5 * abs(b) + 1e-9
    z = torch.zeros(5, 2)
    assert float(l2_loss_nosync(torch.randn(5, 2, generator=gen), z)) == 0.0


CASCADE_OPTS = ["MODEL.FPN_ON", True, "MODEL.CASCADE_ON", True, "MODEL.NUM_CLASSES", 81, "MODEL.CONV1_RGB2BGR", False,
                "MODEL.CLS_AGNOSTIC_BBOX_REG", True, "RPN.ANCHOR_STRIDE", (4, 8, 16, 32, 64),
                "RPN.PRE_NMS_TOP_N_TRAIN", 2000, "RPN.PRE_NMS_TOP_N_TEST", 1000, "RPN.POST_NMS_TOP_N_TEST", 1000,
                "RPN.FPN_POST_NMS_TOP_N_TEST", 1000, "FAST_RCNN.ROI_XFORM_RESOLUTION", (7, 7),
                "FAST_RCNN.ROI_XFORM_SAMPLING_RATIO", 2, "CASCADE_RCNN.NUM_STAGE", 2, "CASCADE_RCNN.TEST_STAGE", 2,
                "CASCADE_RCNN.TEST_ENSEMBLE", True, "CASCADE_RCNN.IOU_HELPER", True,
                "CASCADE_RCNN.IOU_HELPER_MERGE", True, "CASCADE_RCNN.RESCORE_ON", True,
                "CASCADE_RCNN.RESCORE_LOSS_WEIGHT", 0.2, "CASCADE_RCNN.IOU_LOSS_WEIGHT", 1.0]


def test_cascade_rcnn_state_dict_abi():
    """Offset-regression Cascade R-CNN with ISM + RSM (cfgs/rcnn/mscoco/cascade/ISM+RSM): same keys, shapes and
    trainable set as the reference model (tests/golden/model_cascade_meta.json)."""
    from pet.rcnn.core import config
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    with open(os.path.join(os.path.dirname(__file__), "golden", "model_cascade_meta.json")) as f:
        meta = json.load(f)
    config.reset_cfg()
    config.merge_cfg_from_list(CASCADE_OPTS)
    try:
        model = convert_bn2affine_model(Generalized_RCNN(is_train=True))
        got = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        assert got == meta["state_dict"]
        assert [k for k, p in model.named_parameters() if p.requires_grad] == meta["trainable"]
    finally:
        config.reset_cfg()
