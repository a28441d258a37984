"""Process-per-GPU helpers (torch.distributed; backend 'nccl' is RCCL on ROCm, 'gloo' on CPU tests).

Inference shards by image: every rank owns a contiguous slice of the global batch and there is
no collective on the data path -- the process group is only used for the timing barrier, the
max-over-ranks step time and (optionally) gathering the per-image detection lists on rank 0.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world 1)."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend or ('nccl' if torch.cuda.is_available() else 'gloo'),
                                rank=rank, world_size=world)
    return rank, local, world


def shard(total, rank, world):
    """Contiguous image shard [begin, end) of `total` items for `rank` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def barrier(device=None):
    if device is not None and torch.device(device).type == 'cuda':
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
        if device is not None and torch.device(device).type == 'cuda':
            torch.cuda.synchronize(device)


def max_over_ranks(value, device='cpu'):
    """MAX all-reduce of a python float (the slowest rank defines the step time)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device='cpu'):
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_detections(per_image, dst=0):
    """Gather each rank's list of per-image detection lists on `dst` in global image order."""
    if not dist.is_initialized():
        return per_image
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(per_image, out, dst=dst)
    if dist.get_rank() != dst:
        return None
    return [img for part in out for img in part]
