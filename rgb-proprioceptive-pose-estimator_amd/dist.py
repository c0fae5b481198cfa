"""Data-parallel replication: one process per GPU, parameters replicated, per-step SUM all-reduce of the
flat gradient buffer over RCCL (torch.distributed backend "nccl" on ROCm) or gloo on CPU.

Replaces the single-process nn.DataParallel wrappers of the reference (models/naive.py:224,234,253,274).
The loss is a SUM over samples (models/losses.py:75,80,118,122), so the gradient of a global batch is
the SUM of shard gradients: the reduction op is SUM, never AVG.  BatchNorm statistics stay per replica,
as they do under nn.DataParallel.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n, rank, world):
    """Episodes [lo, hi) of rank `rank` (shard along N, never along the LSTM time axis S)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradSync:
    """Bucketed SUM all-reduce over a flat gradient buffer.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of M bytes moves 2*(P-1)/P*M
    through one link, so a few large buckets (default 32 MiB) keep per-call latency negligible while letting
    the first buckets start before the last ones are queued.
    """

    def __init__(self, flat_grad, bucket_bytes=32 << 20, group=None):
        self.flat = flat_grad
        self.group = group
        self.bucket = max(1, bucket_bytes // flat_grad.element_size())
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def buckets(self):
        n = self.flat.numel()
        return [(lo, min(n, lo + self.bucket)) for lo in range(0, n, self.bucket)]

    def all_reduce(self):
        if self.world == 1:
            return
        works = [dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for lo, hi in self.buckets()]
        for w in works:
            w.wait()


def broadcast_parameters(flat_params, buffers=(), src=0, group=None):
    """Make every replica start from rank `src`'s parameters and BN buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.broadcast(flat_params, src, group=group)
    for b in buffers:
        dist.broadcast(b, src, group=group)
