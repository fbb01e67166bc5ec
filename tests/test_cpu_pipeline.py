"""The CPU restatement of a whole CPM R-CNN iteration (oracle/cpu_pipeline.py: the timed body of bench.py's
cpu_baseline legs) at toy size: it runs end to end on the C oracle + torch-CPU, produces the reference's 8 losses and
gradients for every trainable tensor, keeps the configuration's sample caps, and the per-image inference path returns
well-formed detections."""
import numpy as np
import torch


def _model_sd(requires_grad):
    from detfill import det_fill_
    from test_host_logic import CPM_OPTS
    from pet.rcnn.core import config
    from pet.rcnn.modeling.model_builder import Generalized_RCNN
    from pet.utils.net import convert_bn2affine_model
    config.reset_cfg()
    config.merge_cfg_from_list(CPM_OPTS)
    try:
        m = convert_bn2affine_model(Generalized_RCNN(is_train=True))
        det_fill_(m)
        trainable = {k for k, p in m.named_parameters() if p.requires_grad}
        sd = {}
        for k, v in m.state_dict().items():
            t = v.detach().float().clone()
            if k.endswith("fc6.weight") or k.endswith("iou_fc1.weight"):
                t = t.reshape(t.shape[0], -1)
            sd[k] = t.requires_grad_(requires_grad and k in trainable)
        return sd, trainable
    finally:
        config.reset_cfg()


def test_cpu_train_step_and_inference_toy_size():
    from oracle import cpu_pipeline as P
    sd, trainable = _model_sd(True)
    rng = np.random.default_rng(0)
    H, W = 96, 128
    images = torch.from_numpy((rng.uniform(0, 255, (2, 3, H, W)) - 110).astype(np.float32))
    gts = [np.array([[8, 10, 70, 80], [60, 20, 120, 90]], np.float32), np.array([[20, 5, 100, 60]], np.float32)]
    labels = [np.array([3, 17]), np.array([80])]
    losses, counts = P.train_step(sd, images, gts, labels, rng)
    assert set(losses) == {"loss_objectness", "loss_rpn_box_reg", "loss_classifier", "loss_grid_1", "loss_grid_2",
                           "loss_grid_3", "loss_iou_3", "loss_rescore"}
    assert all(np.isfinite(v) for v in losses.values()), losses
    assert 0 < counts["cls"] <= 1024 and 0 < counts["rescore"] <= 1024
    assert 3 <= counts["grid_0"] <= 192 and counts["grid_1"] >= 3 and counts["grid_2"] >= 3      # gts always survive
    got = [k for k in trainable if sd[k].grad is not None and bool(torch.isfinite(sd[k].grad).all())]
    assert len(got) == len(trainable) == 196
    sd2, _ = _model_sd(False)
    b, s, l = P.infer_image(sd2, images[:1])
    assert b.shape[1] == 4 and len(b) == len(s) == len(l)
    if len(b):
        # (scores: NaN where the reference's own `scores ** 0.8`, inference.py:62-76, meets a negative ISM-merged score
        # -- an untrained ISM branch; never infinite)
        assert np.isfinite(b).all() and not np.isinf(s).any() and (l > 0).all() and (l < 81).all()
