"""Loss helpers on the hot path (counterparts of pet/lib/ops/smooth_l1_loss.py:4-28, l2_loss.py:4-11)."""
import os

import torch

_FUSED_CE = os.environ.get("CPM_FUSED_CE", "1") != "0"


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


def l2_loss_nosync(x, target):
    """l2_loss without its nonzero() host round trip, same value up to summation order.  The reference indexes
    x[pos_inds] with the [P, 2] (row, column) pairs of the positive targets, i.e. for every positive entry (r, c) it
    gathers rows r AND c (the column index 0/1 used as a row); with E[i] = 0.5 * sum_j (x[i,j] - t[i,j])^2 that is
    (sum_r cnt[r] * E[r] + n_col0 * E[0] + n_col1 * E[1]) / P."""
    if x.shape[0] < 2 or target.shape[1] != 2:
        return l2_loss(x, target)
    m = (target > 0.0).to(x.dtype)
    e = 0.5 * ((x - target) ** 2).sum(dim=1)
    p = m.sum()
    col = m.sum(dim=0)
    total = (m.sum(dim=1) * e).sum() + col[0] * e[0] + col[1] * e[1]
    return total / p.clamp(min=1.0)


class _L2PairsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, iou, target):
        from . import _hip as H
        H.require_gpu(x, iou, target)
        xc = x.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(xc)
        with H.guard(x.device):
            rc = H.lib().cpm_l2_loss_pairs(H.ptr(xc), H.ptr(iou), H.ptr(target), xc.shape[0], H.ptr(loss), H.ptr(grad),
                                           H.stream())
        H.check(rc, "l2_loss_pairs")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        from . import _hip as H
        return H.scaled_by(grad, g), None, None


def l2_loss_fused(x, target=None, iou=None):
    """l2_loss_nosync as ONE launch for value and gradient (cpm_l2_loss_pairs): x [R, 2] fp32 against target [R, 2],
    or against (1 - iou, iou) when `iou` [R] is given instead."""
    if x.dim() != 2 or x.shape[1] != 2 or x.shape[0] < 2 or x.dtype != torch.float32:
        return l2_loss_nosync(x, target if target is not None else torch.stack([1 - iou, iou], dim=1))
    if target is not None:
        target = target.detach().contiguous()
    if iou is not None:
        iou = iou.detach().contiguous()
    return _L2PairsFn.apply(x, iou, target)


_ce_tickets = {}


class _SoftmaxCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        import ctypes
        from . import _hip as H
        H.require_gpu(logits, labels)
        x = logits.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        rows = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
        key = (x.device.index, H.stream_raw())
        ticket = _ce_tickets.get(key)
        if ticket is None:                   # one per (device, stream): calls that share a ticket must be ordered
            ticket = _ce_tickets[key] = torch.zeros((1,), dtype=torch.int32, device=x.device)
        with H.guard(x.device):
            rc = H.lib().cpm_softmax_ce(H.ptr(x), H.ptr(labels), x.shape[0], x.shape[1], ctypes.c_int64(ignore_index),
                                        H.ptr(loss), H.ptr(grad), H.ptr(rows), H.ptr(ticket), H.stream())
        H.check(rc, "softmax_ce")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        from . import _hip as H
        return H.scaled_by(grad, g), None, None


def cross_entropy_fused(logits, labels, ignore_index=-100):
    """F.cross_entropy(logits, labels) (mean over the rows not labelled ignore_index) as ONE launch for value and
    gradient (cpm_softmax_ce) instead of log_softmax + nll_loss forward and their two backward kernels: the cls and RSM
    heads' losses sit at the very end of the forward pass and the very start of the backward pass, where the device
    waits for the host.  Anything but [R >= 1, C] fp32 logits with int64 labels on the GPU goes to the framework."""
    if (not _FUSED_CE or logits.dim() != 2 or logits.shape[0] < 1 or logits.dtype != torch.float32 or not logits.is_cuda
            or labels.dtype != torch.int64 or labels.dim() != 1 or labels.shape[0] != logits.shape[0]):
        return torch.nn.functional.cross_entropy(logits, labels, ignore_index=ignore_index)
    return _SoftmaxCEFn.apply(logits, labels.detach().contiguous(), int(ignore_index))
