"""Fixed-size pos/neg sampling (counterpart of pet/rcnn/utils/balanced_positive_negative_sampler.py:4-67).

Same distribution as the reference (a uniformly random subset of at most batch*fraction positives, the rest
filled with uniformly random negatives) but without its nonzero()/randperm(numel) host round trips: every
candidate draws a random key, the subset is the top-k of the keys, and the data-dependent counts stay on the
device.  Returns boolean masks (the reference's uint8 masks are deprecated indexing in current torch)."""
import torch


class BalancedPositiveNegativeSampler(object):
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction

    @staticmethod
    def _random_subset(candidates, limit):
        """boolean mask of min(limit, #candidates) uniformly random members of `candidates`, where `limit` is a
        python int or a 0-d device tensor (no synchronisation either way); second value = number selected."""
        n = candidates.numel()
        kmax = min(n, limit if isinstance(limit, int) else n)
        keys = torch.where(candidates, torch.rand(n, device=candidates.device), torch.full((n,), -1.0,
                                                                                            device=candidates.device))
        if isinstance(limit, int):
            top, idx = keys.topk(kmax)
            chosen = top >= 0
        else:
            top, idx = keys.topk(min(n, limit.k_cap))
            chosen = (top >= 0) & (torch.arange(top.numel(), device=keys.device) < limit.value)
        mask = torch.zeros(n, dtype=torch.bool, device=candidates.device)
        mask[idx] = chosen
        return mask, chosen.sum()

    def __call__(self, matched_idxs):
        pos_idx, neg_idx = [], []
        for m in matched_idxs:
            max_pos = int(self.batch_size_per_image * self.positive_fraction)
            pm, num_pos = self._random_subset(m >= 1, max_pos)
            budget = _DeviceLimit(self.batch_size_per_image - num_pos, self.batch_size_per_image)
            nm, _ = self._random_subset(m == 0, budget)
            pos_idx.append(pm)
            neg_idx.append(nm)
        return pos_idx, neg_idx


class _DeviceLimit(object):
    """a device-resident count with a host-known upper bound"""

    def __init__(self, value, k_cap):
        self.value, self.k_cap = value, k_cap
