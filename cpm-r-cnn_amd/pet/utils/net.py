"""Layer factories and BN folding (counterpart of pet/utils/net.py:9-174, hot-path subset)."""
import numpy as np
import torch
import torch.nn as nn

import pet.lib.ops as ops


def make_conv(in_channels, out_channels, kernel=3, stride=1, dilation=1, padding=None, groups=1, use_dwconv=False,
              conv_type="normal", use_bn=False, use_gn=False, use_relu=False, kaiming_init=True, suffix_1x1=False,
              inplace=True, eps=1e-5, gn_group=32):
    """net.py:9-59.  Only the plain conv (+GN +ReLU) variants are on the hot path."""
    if conv_type != "normal" or use_bn or use_dwconv or suffix_1x1:
        raise ValueError("make_conv: only conv_type='normal' without BN / depthwise / 1x1 suffix is supported")
    pad = (dilation * kernel - dilation) // 2 if padding is None else padding
    conv = ops.Conv2d(in_channels, out_channels, kernel_size=kernel, stride=stride, padding=pad, dilation=dilation,
                      groups=groups, bias=not use_gn)
    if kaiming_init:
        nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
    else:
        nn.init.normal_(conv.weight, std=0.01)
    if not use_gn:
        nn.init.constant_(conv.bias, 0)
    mods = [conv]
    if use_gn:
        mods.append(ops.GroupNorm(gn_group, out_channels, eps=eps))
    if use_relu:
        mods.append(ops.ReLU(inplace=inplace))
    return nn.Sequential(*mods) if len(mods) > 1 else conv


def make_fc(dim_in, hidden_dim, use_bn=False, use_gn=False, window=None):
    """net.py:62-74 (plain variant).  `window=(C,H,W)`: the layer reads a flattened feature map (ops.Linear)."""
    if use_bn or use_gn:
        raise ValueError("make_fc: BN/GN variants are outside the hot path")
    fc = ops.Linear(dim_in, hidden_dim, window=window)
    nn.init.kaiming_uniform_(fc.weight, a=1)
    nn.init.constant_(fc.bias, 0)
    return fc


def make_norm(c, norm="bn", eps=1e-5, an_k=10):
    """net.py:77-95.  'bn' is a parameter container until convert_bn2affine_model folds it."""
    if norm == "affine":
        return ops.AffineChannel2d(c)
    if norm == "gn":
        group = 32 if c >= 32 else c
        assert c % group == 0
        return ops.GroupNorm(group, c, eps=eps)
    if norm == "none":
        return None
    if norm in ("an_bn", "an_gn"):
        raise ValueError("mixture norms are outside the hot path")
    return nn.BatchNorm2d(c, eps=eps)


def freeze_params(m):
    for p in m.parameters():
        p.requires_grad = False


def convert_bn2affine_model(module, process_group=None, channel_last=False, merge=True):
    """Replace every BatchNorm by a frozen AffineChannel2d, folding the running statistics
    (gamma/sqrt(var+eps), beta - gamma*mu/sqrt(var+eps)) with the same fp32 numpy arithmetic as the reference
    (net.py:98-130)."""
    mod = module
    if isinstance(module, nn.modules.batchnorm._BatchNorm):
        mod = ops.AffineChannel2d(module.num_features)
        gamma = module.weight.data.detach().cpu().numpy()
        beta = module.bias.data.detach().cpu().numpy()
        if merge:
            mu = module.running_mean.data.detach().cpu().numpy()
            var = module.running_var.data.detach().cpu().numpy()
            inv = np.power(var + module.eps, 0.5)
            gamma, beta = gamma / inv, beta - gamma * mu / inv
        mod.weight.data = torch.from_numpy(np.ascontiguousarray(gamma)).to(module.weight.device)
        mod.bias.data = torch.from_numpy(np.ascontiguousarray(beta)).to(module.weight.device)
        freeze_params(mod)
    for name, child in module.named_children():
        mod.add_module(name, convert_bn2affine_model(child, process_group, channel_last, merge))
    return mod


def mismatch_params_filter(s):
    return [i for i in s if i.split(".")[-1] not in ("num_batches_tracked", "running_mean", "running_var")]


def reduce_tensor(tensor, world_size=1):
    rt = tensor.clone()
    torch.distributed.all_reduce(rt, op=torch.distributed.ReduceOp.SUM)
    rt /= world_size
    return rt
