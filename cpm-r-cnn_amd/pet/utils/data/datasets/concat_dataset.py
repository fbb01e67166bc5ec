"""ConcatDataset that forwards get_img_info (pet/utils/data/datasets/concat_dataset.py:6-23)."""
import bisect

from torch.utils.data.dataset import ConcatDataset as _ConcatDataset


class ConcatDataset(_ConcatDataset):
    def get_idxs(self, idx):
        d = bisect.bisect_right(self.cumulative_sizes, idx)
        return d, (idx if d == 0 else idx - self.cumulative_sizes[d - 1])

    def get_img_info(self, idx):
        d, i = self.get_idxs(idx)
        return self.datasets[d].get_img_info(i)
