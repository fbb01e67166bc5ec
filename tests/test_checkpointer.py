"""CPU: checkpoint format and weight import -- key alignment as dumped from the reference, round trips,
torch.optim.SGD-compatible optimizer state, resume."""
import json
import os

import pytest
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "data_pipeline.json")) as f:
        return json.load(f)


def test_align_and_strip_match_reference(golden):
    from pet.utils.checkpointer import align_and_update_state_dicts, strip_prefix_if_present
    a = golden["align"]
    msd = {k: torch.tensor(-1.0) for k in a["model_keys"]}
    wd = {k: torch.tensor(float(i)) for i, k in enumerate(a["weight_keys"])}
    upd, mismatch = align_and_update_state_dicts(msd, wd, -1)
    assert {k: int(v) for k, v in upd.items()} == a["picked"] and sorted(mismatch) == a["mismatch"]
    for keys, want in golden["strip"]:
        assert list(strip_prefix_if_present({k: 0 for k in keys}, "module.")) == want


class _Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 4, 3, bias=False)
        self.body = nn.Sequential(nn.Conv2d(4, 8, 3), nn.GroupNorm(2, 8), nn.Conv2d(8, 8, 1))
        self.fc = nn.Linear(8, 5)


def _solver():
    from pet.utils.collections import AttrDict
    return AttrDict(OPTIMIZER="SGD", WEIGHT_DECAY=1e-4, WEIGHT_DECAY_GN=0.0, BIAS_DOUBLE_LR=True,
                    BIAS_WEIGHT_DECAY=False, MOMENTUM=0.9, BASE_LR=0.02, MAX_ITER=100, WARM_UP_ITERS=10,
                    WARM_UP_FACTOR=0.1, WARM_UP_METHOD="LINEAR", LR_POLICY="STEP", STEPS=[60, 80], GAMMA=0.1,
                    LR_POW=0.9)


def _reference_style_sgd(model, S):
    """torch.optim.SGD with the three groups of pet/utils/optimizer.py:27-55 (what reference checkpoints hold)."""
    gn = {n + s for n, m in model.named_modules() if isinstance(m, nn.GroupNorm) for s in (".weight", ".bias")}
    w, b, g = [], [], []
    for k, p in model.named_parameters():
        (b if "bias" in k else g if k in gn else w).append(p)
    return torch.optim.SGD([dict(params=w, lr=0.1, weight_decay=S.WEIGHT_DECAY, lr_scale=1),
                            dict(params=b, lr=0.2, weight_decay=0, lr_scale=2),
                            dict(params=g, lr=0.1, weight_decay=0, lr_scale=1)], momentum=S.MOMENTUM)


def test_flat_sgd_state_is_torch_sgd_layout():
    from pet.utils.optimizer import Optimizer
    torch.manual_seed(0)
    S = _solver()
    ref_model = _Tiny()
    ref_opt = _reference_style_sgd(ref_model, S)
    for p in ref_model.parameters():
        p.grad = torch.randn_like(p)
    ref_opt.step()
    saved = ref_opt.state_dict()

    model = _Tiny()
    opt = Optimizer(model, S).build()
    assert opt.state_dict()["state"] == {}                       # nothing before the first step
    opt.load_state_dict(saved)
    assert opt._steps == 1
    mine = opt.state_dict()
    assert [g["params"] for g in mine["param_groups"]] == [g["params"] for g in saved["param_groups"]]
    assert [g["lr"] for g in mine["param_groups"]] == [0.1, 0.2, 0.1]
    assert set(mine["state"]) == set(saved["state"])
    for i in saved["state"]:
        assert torch.equal(mine["state"][i]["momentum_buffer"], saved["state"][i]["momentum_buffer"])
    # and a torch.optim.SGD built the reference way accepts what FlatSGD wrote
    back = _reference_style_sgd(_Tiny(), S)
    back.load_state_dict(mine)
    for i in saved["state"]:
        q = back.param_groups[0]["params"] + back.param_groups[1]["params"] + back.param_groups[2]["params"]
        assert torch.equal(back.state[q[i]]["momentum_buffer"], saved["state"][i]["momentum_buffer"])
    bad = {"state": {}, "param_groups": [dict(g, params=g["params"][:-1]) for g in saved["param_groups"]]}
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)


def test_checkpointer_round_trip_and_resume(tmp_path):
    from pet.utils.checkpointer import CheckPointer, get_weights, load_weights
    from pet.utils.lr_scheduler import LearningRateScheduler
    from pet.utils.optimizer import Optimizer
    torch.manual_seed(1)
    S = _solver()
    ckpt = str(tmp_path / "ckpt")
    # pre-training weights: bare state dict with shorter keys, RGB stem
    pre = {"conv1.weight": torch.randn(4, 3, 3, 3), "body.0.weight": torch.randn(8, 4, 3, 3),
           "module_unrelated": torch.zeros(1)}
    torch.save(pre, str(tmp_path / "pre.pth"))
    model = _Tiny()
    cp = CheckPointer(ckpt, weights_path=str(tmp_path / "pre.pth"), auto_resume=True)
    assert cp.resume is False
    cp.load_model(model, convert_conv1=True)
    assert torch.equal(model.conv1.weight, pre["conv1.weight"][:, [2, 1, 0]])
    assert torch.equal(model.body[0].weight, pre["body.0.weight"])
    assert "fc.weight" in cp.mismatch_keys and "conv1.weight" not in cp.mismatch_keys

    opt = Optimizer(model, S).build()                                # parameters move into the flat buffer
    sched = LearningRateScheduler(opt, S, start_iter=1)
    flat_ptr = opt.flat_param.data_ptr()
    opt.flat_mom.copy_(torch.randn_like(opt.flat_mom))
    opt._steps = 3
    sched.step(37)
    cp.save(model, opt, sched, copy_latest=True, infix="iter")
    assert sorted(os.listdir(ckpt)) == ["model_iter37.pth", "model_latest.pth"]
    assert get_weights(ckpt, "/nonexistent") == os.path.join(ckpt, "model_latest.pth")

    model2 = _Tiny()
    cp2 = CheckPointer(ckpt, weights_path="", auto_resume=True)
    assert cp2.resume is True
    cp2.load_model(model2)
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    opt2 = cp2.load_optimizer(Optimizer(model2, S).build())
    sched2 = cp2.load_scheduler(LearningRateScheduler(opt2, S, start_iter=1))
    assert sched2.iteration == 37 and sched2.info == sched.info and opt2._steps == 1
    for p, q in zip(opt._flat_order, opt2._flat_order):
        i, j = opt._seg_index[id(p)], opt2._seg_index[id(q)]
        b0, e0 = int(opt.seg_begin[i]), int(opt.seg_end[i])
        assert torch.equal(opt.flat_mom[b0:e0], opt2.flat_mom[int(opt2.seg_begin[j]):int(opt2.seg_end[j])])
    # loading into a model whose parameters already live in the flat buffer keeps them there
    load_weights(model, os.path.join(ckpt, "model_latest.pth"))
    assert opt.flat_param.data_ptr() == flat_ptr and model.fc.weight.data_ptr() >= flat_ptr
    # DistributedDataParallel-style prefixes are stripped
    torch.save({"model": {"module." + k: v for k, v in model.state_dict().items()}}, str(tmp_path / "ddp.pth"))
    model3 = _Tiny()
    load_weights(model3, str(tmp_path / "ddp.pth"))
    assert torch.equal(model3.fc.weight, model.fc.weight)


def test_windowed_linear_keeps_reference_state_dict_layout():
    """ops.Linear(window=(C,H,W)) stores its weight as the [K,C,H,W] KRSC filter it is used as, while state_dict /
    load_state_dict / optimizer state keep the reference's [K, C*H*W] tensor."""
    import pet.lib.ops as ops
    from pet.utils.optimizer import Optimizer
    torch.manual_seed(2)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc6 = ops.Linear(5 * 3 * 3, 7, window=(5, 3, 3))
            self.fc7 = ops.Linear(7, 4)
    net = Net()
    assert tuple(net.fc6.weight.shape) == (7, 5, 3, 3)
    assert net.fc6.weight.is_contiguous(memory_format=torch.channels_last)
    sd = net.state_dict()
    assert tuple(sd["fc6.weight"].shape) == (7, 45) and list(sd) == ["fc6.weight", "fc6.bias", "fc7.weight", "fc7.bias"]
    w = torch.randn(7, 45)
    net.load_state_dict({**sd, "fc6.weight": w})
    assert torch.equal(net.fc6.weight.detach().reshape(7, 45), w)
    assert torch.equal(net.state_dict()["fc6.weight"], w)
    # same function as a plain Linear on the flattened NCHW map
    x = torch.randn(6, 5, 3, 3)
    want = torch.nn.functional.linear(x.flatten(1), w, net.fc6.bias)
    got = torch.nn.functional.conv2d(x, net.fc6.weight, net.fc6.bias).flatten(1)
    assert torch.allclose(got, want, atol=1e-5)
    # optimizer state keeps the 2-D layout
    S = _solver()
    ref = nn.Sequential()
    ref.fc6, ref.fc7 = nn.Linear(45, 7), nn.Linear(7, 4)
    ropt = _reference_style_sgd(ref, S)
    for p in ref.parameters():
        p.grad = torch.randn_like(p)
    ropt.step()
    saved = ropt.state_dict()
    opt = Optimizer(net, S).build()
    opt.load_state_dict(saved)
    mine = opt.state_dict()
    for i in saved["state"]:
        assert torch.equal(mine["state"][i]["momentum_buffer"], saved["state"][i]["momentum_buffer"])
    assert tuple(mine["state"][0]["momentum_buffer"].shape) == (7, 45)
