"""Warm-up + step / cosine / poly learning-rate schedule (counterpart of pet/utils/lr_scheduler.py:17-127)."""
from bisect import bisect_right

import numpy as np
from torch.optim.optimizer import Optimizer


class LearningRateScheduler(object):
    def __init__(self, optimizer, solver, start_iter=1, iter_per_epoch=-1, local_rank=0):
        if not isinstance(optimizer, Optimizer):
            raise TypeError("{} is not an Optimizer".format(type(optimizer).__name__))
        assert solver.LR_POLICY in ["STEP", "COSINE", "STEP_COSINE", "POLY"]
        assert solver.WARM_UP_METHOD in ["CONSTANT", "LINEAR"]
        self.optimizer, self.solver = optimizer, solver
        self.base_lr = self.new_lr = solver.BASE_LR
        self.iteration, self.iter_per_epoch, self.local_rank = start_iter, iter_per_epoch, local_rank
        self.max_iter, self.warm_up_iters, self.steps = solver.MAX_ITER, solver.WARM_UP_ITERS, solver.STEPS
        self.info = dict(best_acc=0.0, best_epoch=1, cur_acc=0.0, cur_epoch=1)       # lr_scheduler.py:36 (save_best)

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)

    def get_lr(self):
        S, it = self.solver, self.iteration
        if it <= self.warm_up_iters:
            if S.WARM_UP_METHOD == "CONSTANT":
                f = S.WARM_UP_FACTOR
            else:
                alpha = it / self.warm_up_iters
                f = S.WARM_UP_FACTOR * (1 - alpha) + alpha
            return self.base_lr * f
        if S.LR_POLICY == "STEP":
            return self.base_lr * S.GAMMA ** bisect_right(self.steps, it)
        span = self.max_iter - self.warm_up_iters
        if S.LR_POLICY == "COSINE":
            return 0.5 * self.base_lr * (np.cos((it - self.warm_up_iters - 1) * np.pi / span) + 1.0)
        if S.LR_POLICY == "POLY":
            return self.base_lr * ((1. - float(it - self.warm_up_iters - 1) / span) ** S.LR_POW)
        if it < self.steps[-1]:                                              # STEP_COSINE
            return self.base_lr * S.GAMMA ** bisect_right(self.steps, it)
        base = self.base_lr * S.GAMMA ** bisect_right(self.steps, self.steps[-1] - 1)
        return 0.5 * base * (np.cos((it - self.steps[-1] - 1) * np.pi / (self.max_iter - self.steps[-1])) + 1.0)

    def update_learning_rate(self):
        if self.optimizer.param_groups[0]["lr"] != self.new_lr:
            for g in self.optimizer.param_groups:
                g["lr"] = self.new_lr * g.get("lr_scale", 1)

    def step(self, cur_iter=None):
        self.iteration = self.iteration + 1 if cur_iter is None else cur_iter
        self.new_lr = self.get_lr()
        self.update_learning_rate()
