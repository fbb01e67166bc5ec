"""Loss helpers on the hot path (counterparts of pet/lib/ops/smooth_l1_loss.py:4-28, l2_loss.py:4-11)."""
import torch


def smooth_l1_loss(x, target, beta=1. / 9, reduction="none"):
    n = torch.abs(x - target)
    loss = n if beta < 1e-5 else torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


def l2_loss(x, target):
    """0.5*(x-t)^2 over the rows whose target is > 0 (row selection as l2_loss.py:5: nonzero()[:, 0]),
    divided by the number of selected (row) entries."""
    pos_inds = torch.nonzero(target > 0.0).squeeze(1)
    if pos_inds.shape[0] > 0:
        # pos_inds is [P, 2]: indexing x[pos_inds] gathers ROWS listed in both columns, as the reference does
        cond = torch.abs(x[pos_inds] - target[pos_inds])
        return (0.5 * cond ** 2 / pos_inds.shape[0]).sum()
    return (x * 0.0).sum()
