"""Checkpoint I/O with the reference's file format and class surface (pet/utils/checkpointer.py:13-227).

A checkpoint is `{'model': state_dict, 'optimizer': torch.optim.SGD-format state, 'scheduler': {...}}` written as
`<ckpt>/model_latest.pth` (+ `model_iter<N>.pth`); pre-training weights are a bare state dict.  Files written by the
reference load here and vice versa: state-dict keys are the reference's (tests/test_host_logic.py), and FlatSGD
serialises its flat momentum buffer in torch.optim.SGD's per-parameter layout (pet/utils/optimizer.py).
Loading copies IN PLACE into the model's tensors, so parameters that already live in the flat buffer stay there.
"""
import logging
import os
import shutil
from collections import OrderedDict

import torch

from pet.utils.net import mismatch_params_filter

_log = logging.getLogger("pet.checkpointer")


def _info(msg, local_rank=0):
    if local_rank == 0:
        _log.info(msg)


def _read(path):
    # reference checkpoints carry plain Python / numpy scalars next to the tensors (scheduler info)
    return torch.load(path, map_location=torch.device("cpu"), weights_only=False)


def get_weights(ckpt_path, cfg_test_weights, mode="latest"):
    """TEST.WEIGHTS if that file exists, else <ckpt>/model_<mode>.pth (checkpointer.py:13-18)."""
    if os.path.exists(cfg_test_weights):
        return cfg_test_weights
    return os.path.join(ckpt_path, "model_{}.pth".format(mode))


def strip_prefix_if_present(state_dict, prefix="module."):
    """Drop a DistributedDataParallel `module.` prefix when EVERY key carries it (checkpointer.py:170-187)."""
    if not all(k.startswith(prefix) for k in state_dict.keys()):
        return state_dict
    return OrderedDict((k[len(prefix):], v) for k, v in state_dict.items())


def align_and_update_state_dicts(model_state_dict, weights_dict, local_rank=0):
    """Give every model key the loaded tensor whose key is its LONGEST suffix match ('Conv_Body.layer1.0.conv1.weight'
    takes 'layer1.0.conv1.weight' over 'conv1.weight'); returns (updated dict, model keys nothing matched)
    (checkpointer.py:190-242)."""
    weight_keys = sorted(weights_dict.keys())
    matched = set()
    for mk in sorted(model_state_dict.keys()):
        best = None
        for wk in weight_keys:
            if mk.endswith(wk) and (best is None or len(wk) > len(best)):
                best = wk
        if best is None or len(best) == 0:
            continue
        model_state_dict[mk] = weights_dict[best]
        matched.add(mk)
        _info("{} loaded from {} of shape {}".format(mk, best, tuple(weights_dict[best].shape)), local_rank)
    return model_state_dict, set(model_state_dict.keys()) - matched


def _load_into(model, weights_dict, local_rank):
    sd = model.state_dict()
    sd, mismatch = align_and_update_state_dicts(sd, weights_dict, local_rank)
    model.load_state_dict(sd)
    return mismatch


def load_weights(model, weights_path, local_rank=0):
    """Test-time loading: a full checkpoint or a bare state dict (checkpointer.py:21-33)."""
    blob = _read(weights_path)
    weights = blob["model"] if isinstance(blob, dict) and "model" in blob else blob
    mismatch = _load_into(model, strip_prefix_if_present(weights, "module."), -1)
    _info("The mismatch keys: {}.".format(mismatch_params_filter(sorted(mismatch))), local_rank)
    _info("Loading from weights: {}.".format(weights_path), local_rank)


class CheckPointer(object):
    def __init__(self, ckpt, weights_path=None, auto_resume=True, local_rank=0):
        self.ckpt, self.weights_path, self.auto_resume, self.local_rank = ckpt, weights_path, auto_resume, local_rank
        self.mismatch_keys = set()
        self.resume = self.get_model_latest()
        if self.weights_path:
            self.checkpoint = self._load_file()

    def get_model_latest(self):
        latest = os.path.join(self.ckpt, "model_latest.pth")
        if self.auto_resume and os.path.exists(latest):
            self.weights_path = latest
            return True
        return False

    def _load_file(self):
        return _read(self.weights_path)

    def convert_conv1_rgb2bgr(self, weights_dict):
        """ImageNet weights trained on RGB input feed a BGR pipeline: swap input channels 0 and 2 of the stem
        (checkpointer.py:75-81)."""
        w = weights_dict["conv1.weight"].detach().cpu().clone()
        weights_dict["conv1.weight"] = w[:, [2, 1, 0], :, :].contiguous() if w.shape[1] == 3 else w
        return weights_dict

    def load_model(self, model, convert_conv1=False):
        if self.resume:
            weights = strip_prefix_if_present(self.checkpoint.pop("model"), "module.")
            self.mismatch_keys = _load_into(model, weights, self.local_rank)
            _info("Resuming from weights: {}.".format(self.weights_path), self.local_rank)
        elif self.weights_path:
            weights = strip_prefix_if_present(self.checkpoint, "module.")
            if "vgg16_reducedfc" in self.weights_path:
                raise NotImplementedError("VGG16 name mapping is outside the CPM R-CNN path")
            if convert_conv1:
                weights = self.convert_conv1_rgb2bgr(weights)
            self.mismatch_keys = _load_into(model, weights, self.local_rank)
            _info("Pre-training on weights: {}.".format(self.weights_path), self.local_rank)
        else:
            _info("Training from scratch.", self.local_rank)
        return model

    def load_optimizer(self, optimizer):
        if self.resume:
            optimizer.load_state_dict(self.checkpoint.pop("optimizer"))
        return optimizer

    def load_scheduler(self, scheduler):
        if self.resume:
            scheduler.iteration = self.checkpoint["scheduler"]["iteration"]
            scheduler.info = self.checkpoint["scheduler"]["info"]
        return scheduler

    @staticmethod
    def _blob(model, optimizer, scheduler):
        # (an optimizer update queued beside the next forward pass -- FlatSGD.overlap_next_forward -- must have landed
        # before the parameters are copied out)
        from pet.lib.ops import _hip as _H
        _H.wait_pending_sgd()
        blob = {"model": OrderedDict((k, v.detach().cpu().contiguous()) for k, v in model.state_dict().items())}
        if optimizer is not None:
            blob["optimizer"] = optimizer.state_dict()
        if scheduler is not None:
            blob["scheduler"] = scheduler.state_dict()
        return blob

    def save(self, model, optimizer=None, scheduler=None, copy_latest=True, infix="epoch"):
        os.makedirs(self.ckpt, exist_ok=True)
        latest = os.path.join(self.ckpt, "model_latest.pth")
        tmp = latest + ".tmp"
        torch.save(self._blob(model, optimizer, scheduler), tmp)
        os.replace(tmp, latest)                         # a crash mid-write never leaves a truncated model_latest.pth
        if copy_latest and scheduler:
            shutil.copyfile(latest, os.path.join(self.ckpt, "model_{}{}.pth".format(infix, scheduler.iteration)))
        _info("Saving checkpoint done.", self.local_rank)

    def save_best(self, model, optimizer=None, scheduler=None, remove_old=True, infix="epoch"):
        info = scheduler.info
        if info["cur_acc"] < info["best_acc"]:
            return False
        old = "model_{}{}-{:4.2f}.pth".format(infix, info["best_epoch"], info["best_acc"])
        new = "model_{}{}-{:4.2f}.pth".format(infix, info["cur_epoch"], info["cur_acc"])
        if remove_old and os.path.exists(os.path.join(self.ckpt, old)):
            os.remove(os.path.join(self.ckpt, old))
        info["best_acc"], info["best_epoch"] = info["cur_acc"], info["cur_epoch"]
        os.makedirs(self.ckpt, exist_ok=True)
        torch.save(self._blob(model, optimizer, scheduler), os.path.join(self.ckpt, new))
        shutil.copyfile(os.path.join(self.ckpt, new), os.path.join(self.ckpt, "model_latest.pth"))
        return True
