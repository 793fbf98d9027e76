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


# ---- data-parallel training (BASELINE.json config 4; trainers/train_base_diffusion.py under DDP) ----------
def allreduce_grads_(dist, flat: torch.Tensor, local_n: int, global_n: int) -> torch.Tensor:
    """Turn per-rank gradients of LOCAL-batch-mean losses into the gradient of the GLOBAL-batch-mean loss.

    Each rank's `flat` holds d(mean over its local_n samples)/d(params).  The global mean over
    global_n = sum(local_n) samples is sum_r (local_n_r / global_n) * local_mean_r, so the buffer is scaled by
    local_n / global_n and summed over ranks: ONE all-reduce of the whole flat gradient buffer (RCCL ring
    over xGMI on GPUs, gloo on CPU) -- no per-tensor buckets, the backward writes straight into `flat`."""
    if dist is None:
        return flat
    if local_n != global_n:
        flat.mul_(local_n / float(global_n))
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


class DataParallelTrainer:
    """One optimizer step of trainers/train_base_diffusion.py:23-47 with the image batch sharded over ranks.

    `dm` is a tedm_amd.DiffusionModel on this rank's GPU; every rank holds a full replica, runs the native
    forward+backward on its shard, all-reduces `dm.model.grad_flat` once and applies the same optimizer
    update (so replicas stay bit-identical when the collective is deterministic)."""

    def __init__(self, dm, optimizer, dist=None):
        self.dm, self.opt, self.dist = dm, optimizer, dist

    def step(self, x_local: torch.Tensor, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
             global_batch: Optional[int] = None) -> torch.Tensor:
        n = x_local.shape[0]
        world = self.dist.get_world_size() if self.dist is not None else 1
        gb = global_batch if global_batch is not None else n * world
        loss = self.dm.train_step_native(x_local, t=t, noise=noise)
        allreduce_grads_(self.dist, self.dm.model.grad_flat, n, gb)
        self.opt.step()
        if self.dist is not None:                      # report the global-batch loss like a DDP trainer's log line
            loss = loss * (n / float(gb))
            self.dist.all_reduce(loss, op=self.dist.ReduceOp.SUM)
        return loss
