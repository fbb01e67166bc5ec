"""Index samplers of the data loader (host logic; same sequences as the reference for the same seeds --
tests/test_data_pipeline.py compares with index sequences dumped from the reference's classes).

  DistributedSampler          pet/utils/data/samplers/distributed.py:7-63
  GroupedBatchSampler         pet/utils/data/samplers/grouped_batch_sampler.py:8-114
  IterationBasedBatchSampler  pet/utils/data/samplers/iteration_based_batch_sampler.py:4-30
  RangeSampler                pet/utils/data/samplers/range_sampler.py:5-15
"""
import math

import torch
import torch.distributed as dist
from torch.utils.data.sampler import BatchSampler, Sampler


class DistributedSampler(Sampler):
    """Rank `rank` of `num_replicas` sees one contiguous slice of the (epoch-seeded) permutation, the permutation
    being padded with its own head to a multiple of the world size."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None or rank is None:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("DistributedSampler needs num_replicas/rank or an initialised process group")
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.dataset, self.num_replicas, self.rank, self.shuffle = dataset, num_replicas, rank, shuffle
        self.epoch = 0
        self.num_samples = int(math.ceil(len(dataset) / float(num_replicas)))
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        order = order + order[: self.total_size - n]
        lo = self.num_samples * self.rank
        return iter(order[lo: lo + self.num_samples])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class GroupedBatchSampler(BatchSampler):
    """Mini-batches whose members share a group id (aspect-ratio bucket), each group cut into runs of `batch_size`
    in sampler order, the batches ordered by where their first member appears in the sampler's sequence."""

    def __init__(self, sampler, group_ids, batch_size, drop_uneven=False):
        if not isinstance(sampler, Sampler):
            raise ValueError("sampler should be a torch.utils.data.Sampler, got {}".format(sampler))
        self.sampler = sampler
        self.group_ids = [int(g) for g in torch.as_tensor(group_ids).tolist()]
        self.batch_size = batch_size
        self.drop_uneven = drop_uneven
        self._batches = None
        self._fresh = False

    def _prepare_batches(self):
        per_group = {}
        position = {}
        for pos, idx in enumerate(self.sampler):
            idx = int(idx)
            per_group.setdefault(self.group_ids[idx], []).append(idx)
            position[idx] = pos                       # a repeated index keeps its LAST position, as order[ids] = arange does
        batches = []
        for g in sorted(per_group):
            members = sorted(set(per_group[g]), key=position.__getitem__)
            batches += [members[i: i + self.batch_size] for i in range(0, len(members), self.batch_size)]
        batches.sort(key=lambda b: position[b[0]])
        if self.drop_uneven:
            batches = [b for b in batches if len(b) == self.batch_size]
        return batches

    def __iter__(self):
        if self._fresh:
            self._fresh = False
        else:
            self._batches = self._prepare_batches()
        return iter(self._batches)

    def __len__(self):
        if self._batches is None:
            self._batches = self._prepare_batches()
            self._fresh = True
        return len(self._batches)


class IterationBasedBatchSampler(BatchSampler):
    """Re-runs a batch sampler (new epoch seed each pass) until `num_iterations` batches have been produced,
    counting from `start_iter` (resume)."""

    def __init__(self, batch_sampler, num_iterations, start_iter=0):
        self.batch_sampler = batch_sampler
        self.num_iterations = num_iterations
        self.start_iter = start_iter

    def __iter__(self):
        it = self.start_iter
        while it <= self.num_iterations:
            inner = getattr(self.batch_sampler, "sampler", None)
            if hasattr(inner, "set_epoch"):
                inner.set_epoch(it)
            for batch in self.batch_sampler:
                it += 1
                if it > self.num_iterations:
                    break
                yield batch

    def __len__(self):
        return self.num_iterations


class RangeSampler(Sampler):
    def __init__(self, start_ind, end_ind):
        self.start_ind, self.end_ind = start_ind, end_ind

    def __iter__(self):
        return iter(range(self.start_ind, self.end_ind))

    def __len__(self):
        return self.end_ind - self.start_ind
