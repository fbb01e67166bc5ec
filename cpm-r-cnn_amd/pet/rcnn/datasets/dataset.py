"""Dataset / loader construction (pet/rcnn/datasets/dataset.py:17-134).

Loader workers decode JPEGs and transform the targets; the collated batch carries uint8 pixels and is turned into
the padded fp32 batch on the MI355X by `images.to(device)` (pet/utils/data/collate_batch.py)."""
import bisect
import os

import torch.utils.data

from pet.rcnn.core.config import cfg
from pet.rcnn.datasets.dataset_catalog import contains, get_ann_fn, get_im_dir
from pet.rcnn.datasets.transform import build_transforms
from pet.utils.data import datasets as D
from pet.utils.data import samplers
from pet.utils.data.collate_batch import BatchCollator


def _world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def build_dataset(dataset_list, is_train=True, local_rank=0):
    if not isinstance(dataset_list, (list, tuple)):
        raise RuntimeError("dataset_list should be a list of strings, got {}".format(dataset_list))
    for name in dataset_list:
        assert contains(name), "Unknown dataset name: {}".format(name)
        assert os.path.exists(get_im_dir(name)), "Im dir '{}' not found".format(get_im_dir(name))
    transforms = build_transforms(is_train)
    sets = [D.COCODataset(root=get_im_dir(n), ann_file=get_ann_fn(n), remove_images_without_annotations=is_train,
                          ann_types=("bbox",), transforms=transforms) for n in dataset_list]
    return sets[0] if len(sets) == 1 else D.ConcatDataset(sets)


def make_data_sampler(dataset, shuffle, distributed):
    if distributed:
        if cfg.DATALOADER.SAMPLER_TRAIN != "DistributedSampler":
            raise NotImplementedError("sampler %s is outside the CPM R-CNN path" % cfg.DATALOADER.SAMPLER_TRAIN)
        return samplers.DistributedSampler(dataset, shuffle=shuffle)
    if shuffle:
        return torch.utils.data.sampler.RandomSampler(dataset)
    return torch.utils.data.sampler.SequentialSampler(dataset)


def _aspect_group_ids(dataset, bins):
    bins = sorted(bins)
    ids = []
    for i in range(len(dataset)):
        info = dataset.get_img_info(i)
        ids.append(bisect.bisect_right(bins, float(info["height"]) / float(info["width"])))
    return ids


def make_batch_data_sampler(dataset, sampler, aspect_grouping, images_per_batch, num_iters=None, start_iter=0):
    if aspect_grouping:
        if not isinstance(aspect_grouping, (list, tuple)):
            aspect_grouping = [aspect_grouping]
        batch_sampler = samplers.GroupedBatchSampler(sampler, _aspect_group_ids(dataset, aspect_grouping),
                                                     images_per_batch, drop_uneven=False)
    else:
        batch_sampler = torch.utils.data.sampler.BatchSampler(sampler, images_per_batch, drop_last=False)
    if num_iters is not None:
        batch_sampler = samplers.IterationBasedBatchSampler(batch_sampler, num_iters, start_iter)
    return batch_sampler


def make_train_data_loader(datasets, is_distributed=False, start_iter=0):
    ims_per_gpu = int(cfg.TRAIN.BATCH_SIZE / _world_size())
    aspect_grouping = [1] if cfg.DATALOADER.ASPECT_RATIO_GROUPING else []
    sampler = make_data_sampler(datasets, True, is_distributed)
    batch_sampler = make_batch_data_sampler(datasets, sampler, aspect_grouping, ims_per_gpu, cfg.SOLVER.MAX_ITER,
                                            start_iter)
    return torch.utils.data.DataLoader(datasets, num_workers=cfg.TRAIN.LOADER_THREADS, batch_sampler=batch_sampler,
                                       collate_fn=BatchCollator(cfg.TRAIN.SIZE_DIVISIBILITY))


def make_test_data_loader(datasets, start_ind, end_ind, is_distributed=True):
    if start_ind == -1 or end_ind == -1:
        sampler = samplers.DistributedSampler(datasets) if is_distributed else None
    else:
        sampler = samplers.RangeSampler(start_ind, end_ind)
    return torch.utils.data.DataLoader(datasets, batch_size=cfg.TEST.IMS_PER_GPU, shuffle=False, sampler=sampler,
                                       num_workers=cfg.TEST.LOADER_THREADS,
                                       collate_fn=BatchCollator(cfg.TEST.SIZE_DIVISIBILITY))
