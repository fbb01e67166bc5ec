"""Fixed-size pos/neg sampling (counterpart of pet/rcnn/utils/balanced_positive_negative_sampler.py:4-67).
Returns boolean masks (the reference's uint8 masks are deprecated indexing in current torch)."""
import torch


class BalancedPositiveNegativeSampler(object):
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction

    def __call__(self, matched_idxs):
        pos_idx, neg_idx = [], []
        for m in matched_idxs:
            positive = torch.nonzero(m >= 1).squeeze(1)
            negative = torch.nonzero(m == 0).squeeze(1)
            num_pos = min(positive.numel(), int(self.batch_size_per_image * self.positive_fraction))
            num_neg = min(negative.numel(), self.batch_size_per_image - num_pos)
            perm1 = torch.randperm(positive.numel(), device=positive.device)[:num_pos]
            perm2 = torch.randperm(negative.numel(), device=negative.device)[:num_neg]
            pm = torch.zeros_like(m, dtype=torch.bool)
            nm = torch.zeros_like(m, dtype=torch.bool)
            pm[positive[perm1]] = True
            nm[negative[perm2]] = True
            pos_idx.append(pm)
            neg_idx.append(nm)
        return pos_idx, neg_idx
