#!/usr/bin/env python3
"""Training driver with the reference's command line and flow (tools/rcnn/train_net.py:20-147):

    python tools/rcnn/train_net.py --cfg cfgs/...yaml [KEY VALUE ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/rcnn/train_net.py --cfg ...

cfg -> Generalized_RCNN -> CheckPointer (pre-trained weights or auto-resume) -> frozen-BN fold -> flat-buffer SGD ->
schedule -> COCO loader (uint8 pixels; resize / flip / normalise / pad run on the MI355X inside images.to(device)) ->
loop: forward, backward with the chunked gradient all-reduce overlapped (RCCL over xGMI), one-launch SGD step,
snapshots every SOLVER.SNAPSHOT_ITERS.  One process per GPU; rank and world size come from the launcher's environment.
"""
import argparse
import logging
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cpm-r-cnn_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pet.rcnn.core.config import assert_and_infer_cfg, cfg, merge_cfg_from_file, merge_cfg_from_list  # noqa: E402
from pet.rcnn.datasets import build_dataset, make_train_data_loader  # noqa: E402
from pet.rcnn.modeling.model_builder import Generalized_RCNN  # noqa: E402
from pet.utils.checkpointer import CheckPointer  # noqa: E402
from pet.utils.lr_scheduler import LearningRateScheduler  # noqa: E402
from pet.utils.net import convert_bn2affine_model, mismatch_params_filter  # noqa: E402
from pet.utils.optimizer import Optimizer  # noqa: E402
from pet.utils.parallel import (FlatGradReducer, backward_losses, broadcast_initial_state,  # noqa: E402
                                reduce_losses)

log = logging.getLogger("train_net")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="CPM R-CNN training on MI355X")
    p.add_argument("--cfg", dest="cfg_file", default=None, type=str, help="config file")
    p.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE overrides (pet/rcnn/core/config.py)")
    return p.parse_args(argv)


def train(model, loader, optimizer, scheduler, checkpointer, reducer, device, rank, world):
    model.train()
    t0, seen = time.time(), 0
    for iteration, (images, targets, _) in enumerate(loader, scheduler.iteration):
        scheduler.step()
        optimizer.zero_grad()
        images = images.to(device)                                 # DeferredBatch -> ImageList (cpm_image_prep)
        targets = [t.to(device) for t in targets]
        reducer.begin_step()
        losses = model(images, targets)["losses"]
        backward_losses(losses)                                    # = sum(losses.values()).backward()
        reducer.finish()
        optimizer.step()
        seen += len(targets)
        if scheduler.iteration % cfg.DISPLAY_ITER == 0 or scheduler.iteration == 1:
            shown = reduce_losses(losses) if world > 1 else {k: float(v.detach()) for k, v in losses.items()}
            if rank == 0:
                dt = time.time() - t0
                log.info("iter %d lr %.6f loss %.4f (%s) %.1f img/s/gpu", scheduler.iteration, scheduler.new_lr,
                         sum(shown.values()), ", ".join("%s %.4f" % kv for kv in sorted(shown.items())), seen / dt)
                t0, seen = time.time(), 0
        if rank == 0 and cfg.SOLVER.SNAPSHOT_ITERS > 0 and (iteration + 1) % cfg.SOLVER.SNAPSHOT_ITERS == 0:
            checkpointer.save(model, optimizer, scheduler, copy_latest=True, infix="iter")
    if rank == 0:
        checkpointer.save(model, optimizer, scheduler, copy_latest=True, infix="iter")


def main(argv=None):
    args = parse_args(argv)
    if args.cfg_file:
        merge_cfg_from_file(args.cfg_file)
    if args.opts:
        merge_cfg_from_list(args.opts)
    world = int(os.environ.get("WORLD_SIZE", 1))
    distributed = world > 1
    device = torch.device("cuda", args.local_rank)
    torch.cuda.set_device(device)
    if distributed:
        dist.init_process_group(backend=args.backend, init_method="env://")
    rank = dist.get_rank() if distributed else 0
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(asctime)s %(message)s")
    os.makedirs(cfg.CKPT or ".", exist_ok=True)
    if args.cfg_file and rank == 0:
        shutil.copyfile(args.cfg_file, os.path.join(cfg.CKPT, os.path.basename(args.cfg_file)))
    assert_and_infer_cfg(make_immutable=False)

    model = Generalized_RCNN()
    checkpointer = CheckPointer(cfg.CKPT, weights_path=cfg.TRAIN.WEIGHTS, auto_resume=cfg.TRAIN.AUTO_RESUME,
                                local_rank=rank)
    model = checkpointer.load_model(model, convert_conv1=cfg.MODEL.CONV1_RGB2BGR)
    if cfg.MODEL.BATCH_NORM != "freeze":
        raise ValueError("MODEL.BATCH_NORM must be 'freeze' (every CPM R-CNN config)")
    model = convert_bn2affine_model(model, merge=not checkpointer.resume)
    model = model.to(device).to(memory_format=torch.channels_last)
    optimizer = checkpointer.load_optimizer(Optimizer(model, cfg.SOLVER, local_rank=rank).build())
    # the SGD update streams beside the next iteration's frozen stem / layer1 (FlatSGD.step); checkpoints wait for it
    optimizer.overlap_next_forward = os.environ.get("CPM_SGD_BESIDE_FORWARD", "1") != "0"
    log.info("The mismatch keys: %s", mismatch_params_filter(sorted(checkpointer.mismatch_keys)))
    scheduler = checkpointer.load_scheduler(LearningRateScheduler(optimizer, cfg.SOLVER, start_iter=0, local_rank=rank))
    if distributed:
        # the reference's DistributedDataParallel broadcasts rank 0's parameters and buffers when it wraps the model
        # (train_net.py:134-136); the flat reducer replaces DDP, so the broadcast is explicit here
        broadcast_initial_state(model, optimizer, src=0)
    reducer = FlatGradReducer(optimizer)
    datasets = build_dataset(cfg.TRAIN.DATASETS, is_train=True, local_rank=rank)
    loader = make_train_data_loader(datasets, is_distributed=distributed, start_iter=scheduler.iteration)
    log.info("Training starts: %d images, world size %d, iteration %d -> %d", len(datasets), world, scheduler.iteration,
             cfg.SOLVER.MAX_ITER)
    train(model, loader, optimizer, scheduler, checkpointer, reducer, device, rank, world)
    log.info("Training done.")
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return model


if __name__ == "__main__":
    main()
