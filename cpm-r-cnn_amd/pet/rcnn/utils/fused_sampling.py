"""Batch-wide fixed-size positive/negative sampling without host round trips.

Same distribution as BalancedPositiveNegativeSampler (pet/rcnn/utils/balanced_positive_negative_sampler.py:4-67:
per image a uniformly random subset of at most batch*fraction positives, the remainder filled with uniformly random
negatives), evaluated for ALL images of the batch by one call into the HIP library (cpm_sample_pos_neg: count,
threshold filter, rank-select; pet/lib/ops/detect_glue.py).  No nonzero(), no per-image loop, no sort over the
half million anchors.  There is no host formulation of this path: without the library it raises."""
import pet.lib.ops as ops


def batch_pos_neg_sample(labels, counts, batch_size_per_image, positive_fraction, seed=None):
    """labels [R] (>= 1 positive, 0 negative, < 0 ignored), image-contiguous, `counts` candidates per image (host).
    Returns boolean masks (pos, neg) over the R candidates and the per-image sample sizes [images, 2] (device)."""
    return ops.sample_pos_neg(labels, counts, batch_size_per_image, positive_fraction, seed=seed)
