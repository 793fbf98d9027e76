"""CPU, world_size 2, gloo: the data-parallel plumbing bench.py uses for N > 1 (sharding, barrier,
max-over-ranks timing, whole-job aggregation).  The GPU forward itself needs no collective."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from tedm_amd import dp
    dist = dp.init('gloo')
    assert dp.env_world() == (world, rank, rank)
    beg, end = dp.shard_range(101, world, rank)
    items = torch.arange(101)[beg:end]
    import time

    def step():
        time.sleep(0.01 * (rank + 1))            # rank 1 is slower: the job time is the max over ranks
        return float(items.sum())
    elapsed, last = dp.timed_steps(step, steps=3, warmup=1, dist=dist)
    total = dp.sum_over_ranks(dist, last)
    n = dp.sum_over_ranks(dist, float(end - beg))
    q.put((rank, beg, end, elapsed, total, n))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from tedm_amd import dp
    for n in (0, 1, 7, 32, 101, 256):
        for world in (1, 2, 3, 8):
            r = [dp.shard_range(n, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_timing_and_aggregation():
    world, port = 2, 29631
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, b0, e0, t0, tot0, n0), (r1, b1, e1, t1, tot1, n1) = res
    assert (b0, e0, b1, e1) == (0, 51, 51, 101)
    assert tot0 == tot1 == float(sum(range(101))) and n0 == n1 == 101.0
    assert t0 == t1 and t0 >= 3 * 0.02 * 0.9                # both ranks report the slower rank's time


def _grad_worker(rank, world, port, q):
    """Ragged shards (5 samples over 2 ranks): local-mean gradients from the CPU oracle, combined with the
    product's all-reduce helper, must equal the global-batch gradient."""
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    torch.set_num_threads(2)
    from oracle import unet_ref as O
    from tedm_amd import dp
    from tedm_amd.weights import synth_state_dict
    dist = dp.init('gloo')
    gen = torch.Generator().manual_seed(3)
    N, S, dim = 5, 16, 32
    x0, noise = torch.rand(N, 1, S, S, generator=gen), torch.randn(N, 1, S, S, generator=gen)
    t = torch.randint(0, 1000, (N,), generator=gen)
    bufs = O.diffusion_buffers()
    from tedm_amd.unet import Unet
    m = Unet(dim)                                   # host-side container only: shapes of the 276 keys
    w = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)

    def grads(sl):
        p = {k: v.clone().requires_grad_(True) for k, v in w.items()}
        with torch.enable_grad():
            O.train_loss(p, bufs, x0[sl], t[sl], noise[sl], normalize=True, dim=dim).backward()
        return torch.cat([p[k].grad.reshape(-1) for k in sorted(p)])
    beg, end = dp.shard_range(N, world, rank)
    flat = grads(slice(beg, end))
    dp.allreduce_grads_(dist, flat, end - beg, N)
    full = grads(slice(0, N))
    q.put((rank, float((flat - full).norm() / full.norm())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce_equals_global_batch():
    world, port = 2, 29633
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(err < 1e-5 for _, err in res), res


def test_bucket_plan_covers_the_flat_buffer_in_backward_order():
    from tedm_amd import dp
    from tedm_amd.unet import Unet
    m = Unet(64)
    layout = m.grad_layout()
    assert sorted(layout) == sorted((k, p.numel()) for k, p in m.named_parameters())
    groups = []
    for k, _ in layout:
        if not groups or groups[-1] != dp.top_group(k):
            groups.append(dp.top_group(k))
    assert groups == list(reversed(dp.backward_group_order(4)))            # the buffer completes from its end to its start
    total = sum(n for _, n in layout)
    b = dp.plan_buckets(layout, 4, 4)
    assert len(b) == 4 and b[0][2] == total and b[-1][1] == 0 and b[-1][0] == 'time_mlp'
    assert all(x[1] == y[2] for x, y in zip(b, b[1:]))                     # contiguous slices, closed in completion order
    order = dp.backward_group_order(4)
    off, spans = 0, {}
    for k, n in layout:
        spans[k] = (off, off + n)
        off += n
    for g_last, lo, hi in b:                                               # a closed bucket holds only finished modules
        done = order[:order.index(g_last) + 1]
        for k, (a, e) in spans.items():
            if a >= lo and e <= hi:
                assert dp.top_group(k) in done, (k, g_last)
    assert [x[0] for x in b] == ['ups.1', 'ups.0', 'mid_block1', 'time_mlp']
    assert max(hi - lo for _, lo, hi in b) < 0.45 * total                   # ups.0 (14.6 M of 36.2 M) is the largest indivisible module
    assert len(dp.plan_buckets(layout, 4, 1)) == 1 and 4 < len(dp.plan_buckets(layout, 4, 8)) <= 8    # modules are indivisible


def _bucket_worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from tedm_amd import dp
    from tedm_amd.unet import Unet
    dist = dp.init('gloo')
    m = Unet(32)
    sizes = m.grad_layout()
    total = sum(n for _, n in sizes)
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(total, generator=g)
    local_n, global_n = (3, 5) if rank == 0 else (2, 5)                    # ragged shards: scale by local_n / global_n
    single = dp.allreduce_grads_(dist, local.clone(), local_n, global_n)
    flat = local.clone()
    red = dp.BucketedGradReducer(dist, flat, dp.plan_buckets(sizes, 4, 4), local_n, global_n, timing=True)
    for name in dp.backward_group_order(4):                                 # the sequence the native backward reports
        red.on_group(name)
    red.finish()
    ok = bool(torch.equal(flat, single))
    rep = red.report()                                                      # what one N > 1 driver run has to show
    ok = ok and [b['last_group'] for b in rep['buckets']] == ['ups.1', 'ups.0', 'mid_block1', 'time_mlp']
    ok = ok and sum(b['elements'] for b in rep['buckets']) == total and rep['backward_done_ms'] >= rep['buckets'][-1]['closed_ms']
    ok = ok and all(b['allreduce_done_ms'] >= b['closed_ms'] and b['allreduce_gb_per_s'] > 0 for b in rep['buckets'])
    # second step on the SAME reducer (one side stream / object for the trainer's lifetime): reset, same result
    flat.copy_(local)
    red.reset(local_n, global_n)
    for name in dp.backward_group_order(4):
        red.on_group(name)
    red.finish()
    ok = ok and bool(torch.equal(flat, single))
    q.put((rank, ok, float(flat.double().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_bucketed_allreduce_is_bit_identical_to_single_buffer():
    world, port = 2, 29635
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(eq for _, eq, _ in res), res
    assert res[0][2] == res[1][2]                                           # both ranks hold the same reduced buffer


def _failing_bucket_worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from tedm_amd import dp
    from tedm_amd.unet import Unet
    dist = dp.init('gloo')
    sizes = Unet(32).grad_layout()
    flat = torch.ones(sum(n for _, n in sizes))
    red = dp.BucketedGradReducer(dist, flat, dp.plan_buckets(sizes, 4, 4), 1, 2)
    if rank == 1:                                                            # this rank's callback fails at its second bucket
        real = red._reduce
        def broken(name, lo, hi):
            if name == 'ups.0':
                raise ValueError('injected failure in the backward callback')
            real(name, lo, hi)
        red._reduce = broken
    try:
        for name in dp.backward_group_order(4):
            red.on_group(name)
        red.finish()
        q.put((rank, 'finished'))
    except BaseException as e:
        q.put((rank, type(e).__name__ + ': ' + str(e)[:80]))


def test_a_rank_failing_between_buckets_does_not_leave_its_peer_waiting():
    """an exception inside the backward's mark callback is deferred to finish(); by then the failing rank has issued
    fewer collectives than its peers.  It tears its process group down, so the peer's pending all-reduce FAILS (here: within
    seconds on gloo) instead of waiting for the backend timeout."""
    world, port = 2, 29641
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert 'injected failure' in res[1] or res[1].startswith('RuntimeError'), res
    assert res[0] != 'finished', res                                        # the healthy rank got an error, not a silent hang / success


def _validate_worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from argparse import Namespace
    from tedm_amd import dp, tedm
    dist = dp.init('gloo')
    model = torch.nn.Conv2d(1, 1, 1)
    with torch.no_grad():
        model.weight.fill_(4.0); model.bias.fill_(-2.0)                   # predicts x > 0.5
    g = torch.Generator().manual_seed(7)
    x = torch.rand(4, 1, 8, 8, generator=g)
    y = (x > 0.5).float() if rank == 0 else torch.zeros(4, 1, 8, 8)       # rank 1's shard has NO positives: its recall / Dice are NaN per sample
    if rank == 1:
        x = x * 0.4                                                       # ... and no predicted positives either
    out = tedm.validate_head(Namespace(device='cpu'), model, [(x, y)], dist)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_validation_metrics_are_global_nanmeans_over_ranks():
    """a rank whose validation shard has no positives yields NaN per-sample Dice / recall; the global metric must ignore
    those samples (the reference's nanmean over all samples of one device), not become NaN."""
    world, port = 2, 29647
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_validate_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]                                               # every rank reports the global value
    for k in ('val/loss', 'val/dice', 'val/recall', 'val/dice_ensemble'):
        assert res[0][k] == res[0][k], (k, res[0])                        # not NaN
    assert res[0]['val/dice'] == 1.0 and res[0]['val/recall'] == 1.0      # rank 0's four perfect samples; rank 1's NaN samples ignored
