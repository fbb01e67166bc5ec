from .samplers import DistributedSampler, GroupedBatchSampler, IterationBasedBatchSampler, RangeSampler
