"""pet.lib.utils.comm.get_world_size (pet/lib/utils/comm.py:16-30)."""
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
