"""Batch-wide fixed-size positive/negative sampling without host round trips.

Same distribution as BalancedPositiveNegativeSampler (pet/rcnn/utils/balanced_positive_negative_sampler.py:4-67:
per image a uniformly random subset of at most batch*fraction positives, the remainder filled with uniformly random
negatives), evaluated for ALL images of the batch at once: every candidate draws a random key inside its
(image, class) bucket, one sort orders the buckets, and a candidate is taken when its rank within the bucket is
below the bucket's device-resident quota.  No nonzero(), no per-image loop."""
import torch


def batch_pos_neg_sample(labels, img, n_img, batch_size_per_image, positive_fraction):
    """labels [R] (>= 1 positive, 0 negative, < 0 ignored), img [R] int image index.
    Returns boolean masks (pos, neg) over the R candidates."""
    R = labels.numel()
    dev = labels.device
    img = img.long()
    cls = torch.where(labels >= 1, 0, torch.where(labels == 0, 1, 2))
    bucket = img * 3 + cls                                             # [R] in [0, 3*n_img)
    key = bucket.to(torch.float32) + torch.rand(R, device=dev) * 0.998
    order = torch.argsort(key)
    b_sorted = bucket[order]
    # bucket boundaries in the sorted order (a binary search per bucket; an index_add_ of R ones into 3*n_img bins is
    # R atomics on a handful of addresses)
    bounds = torch.searchsorted(b_sorted, torch.arange(3 * n_img + 1, device=dev))
    starts, counts = bounds[:-1], bounds[1:] - bounds[:-1]
    c = counts.view(n_img, 3)
    max_pos = int(batch_size_per_image * positive_fraction)
    n_pos = c[:, 0].clamp(max=max_pos)
    n_neg = torch.minimum(c[:, 1], batch_size_per_image - n_pos)
    quota = torch.stack([n_pos, n_neg, torch.zeros_like(n_pos)], dim=1).view(-1)
    rank = torch.arange(R, device=dev) - starts[b_sorted]
    take = torch.empty(R, dtype=torch.bool, device=dev)
    take[order] = rank < quota[b_sorted]
    return take & (cls == 0), take & (cls == 1)
