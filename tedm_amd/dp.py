"""Data-parallel plumbing for the denoise / feature-extraction path.

The UNet forward has no cross-sample operation (GroupNorm per sample, LayerNorm per pixel, attention
per (sample, head); no BatchNorm), so inference shards the image batch across ranks with NO data-path
collective: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm, "gloo" on CPU
for tests) only carries the barrier and the max-over-ranks of the elapsed time.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch


def env_world() -> Tuple[int, int, int]:
    """(world_size, rank, local_rank) from the torchrun environment (1, 0, 0 when absent)."""
    return (int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')),
            int(os.environ.get('LOCAL_RANK', '0')))


def init(backend: Optional[str] = None):
    """Initialise the default process group when WORLD_SIZE > 1. Returns torch.distributed or None."""
    world, rank, local = env_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    kw = {}
    if backend == 'nccl':
        torch.cuda.set_device(local)
        kw['device_id'] = torch.device('cuda', local)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) shard of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    beg = rank * base + min(rank, rem)
    return beg, beg + base + (1 if rank < rem else 0)


def barrier(dist, device: Optional[torch.device] = None) -> None:
    if device is not None and device.type == 'cuda':
        torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
        if device is not None and device.type == 'cuda':
            torch.cuda.synchronize(device)


def max_over_ranks(dist, value: float, device: Optional[torch.device] = None) -> float:
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None and device.type == 'cuda' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value: float, device: Optional[torch.device] = None) -> float:
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None and device.type == 'cuda' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def timed_steps(step, steps: int, warmup: int, dist=None, device: Optional[torch.device] = None):
    """bench.py's timing contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by a
    barrier + device synchronize on both sides; returns (elapsed seconds = max over ranks, last result)."""
    import time
    out = None
    for _ in range(warmup):
        out = step()
    barrier(dist, device)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier(dist, device)
    return max_over_ranks(dist, time.perf_counter() - t0, device), out
