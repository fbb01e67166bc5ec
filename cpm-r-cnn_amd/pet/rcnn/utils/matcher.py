"""IoU matcher (counterpart of pet/rcnn/utils/matcher.py:4-111): argmax over gts per prediction, two
thresholds (-1 below low, -2 between), optional low-quality matches for every gt's best predictions."""
import torch


class Matcher(object):
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False):
        assert low_threshold <= high_threshold
        self.high_threshold = high_threshold
        self.low_threshold = low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix):
        if match_quality_matrix.numel() == 0:
            if match_quality_matrix.shape[0] == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
            raise ValueError("No proposal boxes available for one of the images during training")
        vals, matches = match_quality_matrix.max(dim=0)
        best = matches.clone() if self.allow_low_quality_matches else None
        # torch.where instead of masked assignment: `t[mask] = v` runs a nonzero and synchronises with the device
        matches = torch.where(vals < self.low_threshold, torch.full_like(matches, Matcher.BELOW_LOW_THRESHOLD), matches)
        matches = torch.where((vals >= self.low_threshold) & (vals < self.high_threshold),
                              torch.full_like(matches, Matcher.BETWEEN_THRESHOLDS), matches)
        if self.allow_low_quality_matches:
            row_max, _ = match_quality_matrix.max(dim=1)
            tied = (match_quality_matrix == row_max[:, None]).any(dim=0)     # predictions tying some gt's best
            matches = torch.where(tied, best, matches)
        return matches
