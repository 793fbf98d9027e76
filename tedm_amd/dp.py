"""Data-parallel plumbing for the denoise / feature-extraction path.

The UNet forward has no cross-sample operation (GroupNorm per sample, LayerNorm per pixel, attention
per (sample, head); no BatchNorm), so inference shards the image batch across ranks with NO data-path
collective: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm, "gloo" on CPU
for tests) only carries the barrier and the max-over-ranks of the elapsed time.
"""
from __future__ import annotations

import os
import time
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
    over xGMI on GPUs, gloo on CPU), the backward writes straight into `flat`.  `BucketedGradReducer` below does the same
    in ~4 slices launched while the backward is still running."""
    if dist is None:
        return flat
    if local_n != global_n:
        flat.mul_(local_n / float(global_n))
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


# Backward completion order of the top-level modules (include/tedm_hip.h: tedm_unet_set_backward_callback); the flat gradient
# buffer is laid out in PARAMETER order, so it completes from its end towards its start (init_conv / time_mlp last).
def backward_group_order(n_levels: int = 4):
    return (['final_conv', 'final_res_block'] + [f'ups.{i}' for i in reversed(range(n_levels))] +
            ['mid_block2', 'mid_attn', 'mid_block1'] + [f'downs.{i}' for i in reversed(range(n_levels))] + ['init_conv', 'time_mlp'])


def top_group(key: str) -> str:
    """Top-level module of a state-dict key: 'ups.1.0.block1.proj.weight' -> 'ups.1', 'mid_attn.fn...' -> 'mid_attn'."""
    parts = key.split('.')
    return '.'.join(parts[:2]) if parts[0] in ('ups', 'downs') else parts[0]


def grad_layout(named_sizes, n_levels: int = 4):
    """Order of the parameters inside the flat gradient buffer: the REVERSE of the backward's completion order of their
    top-level modules (time_mlp, init_conv, downs.0.., mid_block1, mid_attn, mid_block2, ups.0.., final_res_block, final_conv), so
    that the buffer becomes final contiguously from its end towards its start while the backward runs.  Within a module the
    `named_parameters()` order is kept.  `named_sizes`: iterable of (key, numel)."""
    named_sizes = list(named_sizes)
    rank = {g: i for i, g in enumerate(reversed(backward_group_order(n_levels)))}
    unknown = sorted({top_group(k) for k, _ in named_sizes} - set(rank))
    if unknown:
        raise ValueError(f'parameters outside the known top-level modules: {unknown}')
    return sorted(named_sizes, key=lambda kn: rank[top_group(kn[0])])          # stable: keeps the order inside a module


def plan_buckets(layout, n_levels: int = 4, n_buckets: int = 4):
    """Split the flat gradient buffer (parameters in `grad_layout` order) into ~n_buckets contiguous slices that become
    complete, one after the other, while the backward is still running.

    Returns [(last_group, lo, hi)] in completion order: when the backward reports `last_group` done, flat[lo:hi] is final.
    Boundaries sit on top-level modules and balance BYTES (default model, 4 buckets: [final .. ups.1] 5.4 M, ups.0 14.6 M,
    mid 10.2 M, downs + stem + time MLP 6.0 M elements).  The last bucket always closes with 'time_mlp' (written last)."""
    order = backward_group_order(n_levels)
    off, spans = 0, {}
    for k, n in layout:
        g = top_group(k)
        lo, hi = spans.get(g, (off, off))
        if hi != off:
            raise ValueError(f'layout is not grouped by top-level module at {k}')
        spans[g] = (lo, off + n)
        off += n
    total = off
    if [g for g in order if g in spans] != list(reversed(list(spans))):
        raise ValueError('layout is not the reverse of the backward completion order (use dp.grad_layout)')
    target = total / float(max(1, n_buckets))
    buckets, acc, hi = [], 0, total
    present = [g for g in order if g in spans]
    prev = None
    for g in present[:-1]:
        lo_g, hi_g = spans[g]
        size = hi_g - lo_g
        if prev is not None and acc > 0 and acc + size - target > target - acc and len(buckets) < n_buckets - 1:
            buckets.append((prev, hi_g, hi))            # closing BEFORE g leaves the bucket nearer the target than taking g in
            hi, acc = hi_g, 0
        acc += size
        if acc >= target and len(buckets) < n_buckets - 1:
            buckets.append((g, lo_g, hi))
            hi, acc = lo_g, 0
        prev = g
    buckets.append((present[-1], 0, hi))
    assert buckets[0][2] == total and all(a[1] == b[2] for a, b in zip(buckets, buckets[1:]))
    return buckets


class BucketedGradReducer:
    """sum-all-reduce the flat gradient buffer in ~4 buckets, each launched as soon as the backward has produced it
    (decoder gradients first), so that RCCL traffic over xGMI hides under the remaining backward kernels.

    `on_group(name)` is the backward's mark callback: at a bucket's last group it records an event on the compute stream and
    enqueues, on a side stream that waits for that event, the scale by local_n/global_n and the all-reduce of the slice.
    `finish()` makes the compute stream wait for all of them.  Bucketing does not change the result: every element is scaled
    and summed over ranks exactly as by the single-buffer `allreduce_grads_` (bit-identical on 2 ranks, tests/test_dp_gloo.py).
    Works on CPU tensors (gloo) for tests: no streams, the collectives run inline."""

    def __init__(self, dist, flat: torch.Tensor, buckets, local_n: int, global_n: int, timing: bool = False):
        self.dist, self.flat, self.buckets = dist, flat, list(buckets)
        self.by_group = {g: (lo, hi) for g, lo, hi in self.buckets}
        self.cuda = flat.is_cuda
        self.side = torch.cuda.Stream(flat.device) if self.cuda else None        # ONE side stream for the reducer's lifetime
        self.timing = timing
        self.pending, self.done, self.marks, self.failed = [], [], [], None
        self.t0 = None
        self.reset(local_n, global_n)

    def reset(self, local_n: int, global_n: int) -> None:
        """Start of a step (before the training forward): the reducer and its stream are reused, only the per-step state resets."""
        self.scale = None if local_n == global_n else local_n / float(global_n)
        self.pending, self.done, self.marks, self.failed = [], [], [], None
        if self.timing:
            if self.cuda:
                self.t0 = torch.cuda.Event(enable_timing=True)
                self.t0.record(torch.cuda.current_stream(self.flat.device))
            else:
                self.t0 = time.perf_counter()

    def on_group(self, name: str) -> None:
        span = self.by_group.get(name)
        if span is None or self.dist is None or self.failed is not None:
            return
        try:
            self._reduce(name, *span)
        except BaseException as e:        # raised inside the C backward's callback: remember it, tell the peers, re-raise in finish()
            self.failed = e
            self._abort_peers()

    def _reduce(self, name: str, lo: int, hi: int) -> None:
        sl = self.flat[lo:hi]
        self.done.append(name)
        if not self.cuda:
            t_closed = time.perf_counter()
            if self.scale is not None:
                sl.mul_(self.scale)
            self.dist.all_reduce(sl, op=self.dist.ReduceOp.SUM)
            if self.timing:
                self.marks.append((name, hi - lo, t_closed, time.perf_counter()))
            return
        ev = torch.cuda.Event(enable_timing=self.timing)
        ev.record(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            if self.scale is not None:
                sl.mul_(self.scale)
            work = self.dist.all_reduce(sl, op=self.dist.ReduceOp.SUM, async_op=True)
            self.pending.append(work)
            if self.timing:
                work.wait()                                  # (side stream only: the compute stream keeps running the backward)
                done = torch.cuda.Event(enable_timing=True)
                done.record(self.side)
                self.marks.append((name, hi - lo, ev, done))

    def _abort_peers(self) -> None:
        """A rank that fails between two buckets has issued fewer collectives than its peers, which would wait for it until the
        backend's timeout.  Tear the process group down so that their pending collective fails now."""
        try:
            import torch.distributed as tdist
            pg = tdist.distributed_c10d._get_default_group()
            for meth in ('abort', '_abort', '_shutdown'):
                be = None
                try:
                    be = pg._get_backend(self.flat.device)
                except Exception:
                    pass
                if be is not None and hasattr(be, meth):
                    getattr(be, meth)()
                    return
            tdist.destroy_process_group()
        except Exception:
            pass

    def finish(self) -> None:
        if self.failed is not None:
            raise RuntimeError(f'bucketed all-reduce failed after the groups {self.done}; the process group was torn down so '
                               f'that the other ranks fail instead of waiting') from self.failed
        if len(self.done) != len(self.buckets) and self.dist is not None:
            raise RuntimeError(f'bucketed all-reduce: backward reported {self.done}, expected the groups {[b[0] for b in self.buckets]}')
        for w in self.pending:
            w.wait()                                         # the current stream waits for the collective
        if self.cuda:
            cur = torch.cuda.current_stream(self.flat.device)
            if self.timing:
                self.t_bwd = torch.cuda.Event(enable_timing=True)
                self.t_bwd.record(cur)                       # every backward kernel is enqueued in front of this
            cur.wait_stream(self.side)
        elif self.timing:
            self.t_bwd = time.perf_counter()
        self.pending = []

    def report(self) -> dict:
        """Per-bucket event times of the last step (timing=True), milliseconds since `reset()`: when the backward had produced
        the bucket, when its all-reduce had finished, when the whole backward had finished; all-reduce GB/s = payload bytes over
        (done - max(closed, previous done)).  Synchronises the device."""
        if not self.timing or self.t0 is None:
            return {}
        out, prev_done = [], 0.0
        if self.cuda:
            torch.cuda.synchronize(self.flat.device)
            ms = lambda e: self.t0.elapsed_time(e)
        else:
            ms = lambda t: (t - self.t0) * 1e3
        for name, n, closed, done in self.marks:
            c, d = ms(closed), ms(done)
            busy = max(d - max(c, prev_done), 1e-6)
            out.append({'last_group': name, 'elements': int(n), 'closed_ms': round(c, 3), 'allreduce_done_ms': round(d, 3),
                        'allreduce_gb_per_s': round(4.0 * n / busy / 1e6, 2)})
            prev_done = d
        bwd = ms(self.t_bwd) if getattr(self, 't_bwd', None) is not None else None
        return {'buckets': out, 'backward_done_ms': None if bwd is None else round(bwd, 3),
                'hidden_under_backward': bool(out) and bwd is not None and all(b['allreduce_done_ms'] <= bwd for b in out[:-1])}


class DataParallelTrainer:
    """One optimizer step of trainers/train_base_diffusion.py:23-47 with the image batch sharded over ranks.

    `dm` is a tedm_amd.DiffusionModel on this rank's GPU; every rank holds a full replica, runs the native
    forward+backward on its shard, all-reduces `dm.model.grad_flat` once and applies the same optimizer
    update (so replicas stay bit-identical when the collective is deterministic)."""

    def __init__(self, dm, optimizer, dist=None, n_buckets: int = 4, timing: bool = False):
        self.dm, self.opt, self.dist, self.n_buckets, self.timing = dm, optimizer, dist, n_buckets, timing
        self._buckets = None
        self._reducer: Optional[BucketedGradReducer] = None     # built once: its side stream lives as long as the trainer

    def step(self, x_local: torch.Tensor, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
             global_batch: Optional[int] = None) -> torch.Tensor:
        n = x_local.shape[0]
        world = self.dist.get_world_size() if self.dist is not None else 1
        gb = global_batch if global_batch is not None else n * world
        unet = self.dm.model
        if self.dist is None or self.n_buckets <= 1:
            loss = self.dm.train_step_native(x_local, t=t, noise=noise)
            allreduce_grads_(self.dist, unet.grad_flat, n, gb)
        else:                                           # buckets go out while the backward is still running
            if self._buckets is None:
                self._buckets = plan_buckets(unet.grad_layout(), len(unet._cfg['dim_mults']), self.n_buckets)
            unet.grad_views(x_local.device)              # allocates grad_flat on first use
            red = self._reducer
            if red is None or red.flat.data_ptr() != unet.grad_flat.data_ptr():
                red = self._reducer = BucketedGradReducer(self.dist, unet.grad_flat, self._buckets, n, gb, timing=self.timing)
            else:
                red.reset(n, gb)
            loss = self.dm.train_step_native(x_local, t=t, noise=noise, on_backward_group=red.on_group)
            red.finish()
        self.opt.step()
        if self.dist is not None:                      # report the global-batch loss like a DDP trainer's log line
            loss = loss * (n / float(gb))
            self.dist.all_reduce(loss, op=self.dist.ReduceOp.SUM)
        return loss

    def comm_report(self) -> dict:
        """What one driver run of N > 1 ranks should show: the backend and world size torch.distributed
        reports, and -- with timing=True -- the bucket / all-reduce / backward event times of the last step."""
        rep = {'dist_backend': self.dist.get_backend() if self.dist is not None else None,
               'world_size': self.dist.get_world_size() if self.dist is not None else 1, 'grad_buckets': self.n_buckets}
        if self._reducer is not None:
            rep.update(self._reducer.report())
        return rep
