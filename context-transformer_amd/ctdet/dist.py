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


class GradBucketer:
    """Gradient all-reduce for the training path (RCCL over xGMI on the GPUs, gloo in CPU tests).

    The reference trains with single-process `torch.nn.DataParallel` (train.py:296-297), i.e. an
    implicit reduce of every gradient to GPU 0.  Here each GPU runs its own process; gradients are
    written into ONE flat fp32 buffer laid out in the order the backward pass produces them (heads
    first, conv1_1 last) and every `bucket_bytes` slice is all-reduced asynchronously as soon as its
    last gradient exists, so the collective of the late layers overlaps the backward kernels of the
    early ones.  xGMI is point-to-point (7 links/GPU), so buckets are large (default 32 MiB): a few
    big ring steps per link instead of hundreds of small ones.
    """

    def __init__(self, numels, device, bucket_bytes=32 << 20, group=None, flat=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.offsets = [0]
        for n in numels:
            self.offsets.append(self.offsets[-1] + int(n))
        # `flat`: the caller's own gradient buffer (the training engine's arena) instead of a private staging one
        self.flat = flat if flat is not None else torch.zeros(self.offsets[-1], dtype=torch.float32, device=device)
        assert self.flat.numel() >= self.offsets[-1] and self.flat.dtype == torch.float32
        per = max(1, bucket_bytes // 4)
        self.bucket_of = []              # production index -> bucket id
        self.bucket_span = []            # bucket id -> (first elem, last elem)
        start, b = 0, 0
        for i in range(len(numels)):
            self.bucket_of.append(b)
            if self.offsets[i + 1] - start >= per or i == len(numels) - 1:
                self.bucket_span.append((start, self.offsets[i + 1]))
                start, b = self.offsets[i + 1], b + 1
        self.last_in_bucket = {}
        for i, bid in enumerate(self.bucket_of):
            self.last_in_bucket[bid] = i
        self.handles = []

    def view(self, i):
        return self.flat[self.offsets[i]:self.offsets[i + 1]]

    def begin(self):
        self.handles = []

    def ready(self, i, before_launch=None):
        """Gradient i (in production order) has been written to view(i).  `before_launch()` runs right before a
        bucket's all-reduce is issued (the training engine joins its weight-gradient stream there)."""
        bid = self.bucket_of[i]
        if self.last_in_bucket[bid] == i and before_launch is not None:
            before_launch()
        if self.world > 1 and self.last_in_bucket[bid] == i:
            a, b = self.bucket_span[bid]
            self.handles.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True))

    def finish(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.world > 1:
            self.flat.div_(self.world)


def global_normalizer(n_local, device):
    """Loss normaliser for data-parallel training: the reference divides by N = sum of positives
    over the WHOLE batch (multibox_loss_combined.py:119-122 on DataParallel-gathered outputs).  With
    per-rank losses and mean-reduced gradients that is N_global / world on every rank."""
    if not dist.is_initialized():
        return n_local
    t = n_local.detach().to(device=device, dtype=torch.float64).reshape(1).clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    # a float quotient: 7 positives over 2 ranks is 3.5, not 3 (an integer cast would change the loss scale)
    out = t / dist.get_world_size()
    return (out if not n_local.dtype.is_floating_point else out.to(n_local.dtype)).reshape(n_local.shape)
