#!/usr/bin/env python3
"""bench.py -- denoise-step throughput of the MI355X-native TEDM UNet (BASELINE.json metric).

A "step" = one pass of the hot path over one synthetic batch that is already resident in HBM:
q-sample (x0, t, noise -> x_t) followed by the UNet noise-prediction forward, reference defaults
(Unet(dim=64, dim_mults=(1,2,4,8), channels=1), 36.2 M parameters), batch 32 x 1 x 256 x 256 per GPU
(BASELINE configs[1]).  Weights are the build's deterministic synthetic generator (no trained TEDM
checkpoint ships with the reference), data is synthetic.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, the batch of images is sharded across ranks (weak scaling, 32 images per
GPU, no data-path collective -- the forward has no cross-sample op); RCCL is used only for the
barrier and the max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from argparse import Namespace
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0         # HBM3E spec, same guide (6.29 TB/s measured copy)
A_FLOPS_PER_IMG_256 = 236.27e9   # SURVEY.md 8d (scales with S^2 up to the 0.5 GF attention term)
A_BYTES_PER_IMG_256 = 939.3e6    # SURVEY.md 8d barrier model, bf16 activations (this build stores fp32: 2x)
T_ROOF_PER_IMG_256 = 159.24e-6   # SURVEY.md App. B stage-wise roofline at nominal peaks, bf16 storage


def host_cores() -> int:
    """CPUs this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = Path(path).read_text().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(Path('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read_text())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def pmc_traffic(kernel: str):
    """HBM-side bytes per launch of `kernel` from the newest committed PMC summary (tools/pmc_traffic.py:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command), or None."""
    return pmc_traffic_with_source(kernel)[0]


def pmc_traffic_with_source(kernel: str):
    """(bytes per launch, name of the profiles/ file they come from) -- the bench line names its source"""
    files = sorted((ROOT / 'profiles').glob('*pmc_traffic.json'))
    if not files:
        return None, None
    try:
        ks = json.loads(files[-1].read_text())['kernels']
    except Exception:
        return None, None
    # rocprof prints every template argument; the library's label omits the trailing QIO = 0 of conv_igemm_kernel
    for name in (kernel, kernel[:-1] + ',0>' if kernel.endswith('>') else kernel):
        if name in ks:
            return ks[name].get('hbm_bytes_per_launch'), files[-1].name
    return None, files[-1].name


def bench_extract(args, dm, dp, dist, dev, world, rank, x0, noise):
    """BASELINE config 3: multi-timestep feature extraction (models/datasetDM_model.py:67-83), steps
    [50,200,400,600,800] (config.py:78), B images per GPU -> (B, 960*5, S, S) fp32 on the device."""
    import torch
    from tedm_amd import DatasetDM
    steps = [50, 200, 400, 600, 800]
    ddm = DatasetDM(Namespace(normalize=True, dim=64, saved_diffusion_model='/nonexistent', verbose=False, device='cuda',
                              t_steps_to_save=steps, max_images_per_forward=64))
    ddm.diffusion_model = dm
    ddm = ddm.to(dev).eval()
    B, S = x0.shape[0], x0.shape[-1]

    out = torch.empty(B, ddm.feature_channels * len(steps), S, S, dtype=torch.float32, device=dev)   # 40 GB at B=32

    def step():
        return ddm.extract_features(x0, noise=noise, out=out)
    elapsed, feats = dp.timed_steps(step, args.steps, args.warmup, dist, dev)
    if rank == 0:
        print(json.dumps({
            'metric': 'feature-extract images/sec @256x256, 5 timesteps (DatasetDM.extract_features)',
            'value': round(world * B * args.steps / elapsed, 2), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.prec, 'data': 'synthetic',
            'config': {'workload': f'DatasetDM.extract_features, {B}x1x{S}x{S} per GPU, t_steps={steps} (BASELINE configs[2])',
                       'feature_tensor': list(feats.shape), 'unet_forwards_per_image': len(steps)}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_train(args, dm, dp, dist, dev, world, rank, x0, t, noise):
    """BASELINE config 4: one optimizer step of trainers/train_base_diffusion.py:36-47 per GPU-local batch
    (q-sample, recorded UNet forward, L1/p2 loss, full backward as HIP kernels), the all-reduce of the flat gradient
    buffer (RCCL) in `--buckets` slices launched on a side stream as the backward completes them, and a fused Adam update.
    Weak scaling: B images per GPU."""
    import torch
    dm.train()
    opt = torch.optim.Adam(dm.model.parameters(), lr=1e-4, fused=True)
    tr = dp.DataParallelTrainer(dm, opt, dist, n_buckets=args.buckets)
    B, S = x0.shape[0], x0.shape[-1]

    def step():
        return tr.step(x0, t=t, noise=noise)
    elapsed, loss = dp.timed_steps(step, args.steps, args.warmup, dist, dev)
    # one more step OUTSIDE the timed region with event pairs around every bucket: when the backward had produced it, when its
    # all-reduce had finished, when the backward had finished, all-reduce GB/s (the overlap evidence of an N > 1 run)
    tr.timing = True
    tr._reducer = None
    tr.step(x0, t=t, noise=noise)
    comm = tr.comm_report()
    if rank == 0:
        print(json.dumps({
            'metric': 'train-step images/sec @256x256 (fwd + bwd + grad all-reduce + Adam)',
            'value': round(world * B * args.steps / elapsed, 2), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.prec, 'data': 'synthetic',
            'config': {'workload': f'DiffusionModel.train_step + backward + Adam, {B}x1x{S}x{S} per GPU, global batch {B * world} '
                                   f'(BASELINE configs[3])', 'final_loss': round(float(loss), 6), 'grad_allreduce_buckets': args.buckets if dist is not None else 0,
                       'gradient_parity_vs_fp32_autograd': {'f16f4': '<= 3e-3 per tensor, 5e-4 over all parameters', 'bf16x3': '<= 1e-4 per tensor, 3e-5 overall'}.get(args.prec, 'n/a'),
                       'train_workspace_GiB': round(dm.model._train_ws.numel() / 2 ** 30, 2)},
            'dist_backend': comm.get('dist_backend'), 'world_size': comm.get('world_size'), 'grad_allreduce': comm}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_tedm(args, dm, dp, dist, dev, world, rank):
    """BASELINE config 5 shape: TEDM = frozen diffusion UNet + SHARED-WEIGHTS pixel classifier (trainers/train_datasetDM.py:30-42)
    trained with BCE on replicated targets (trainers/train_baseline.py:30,44), JSRT-shaped synthetic data (B,1,128,128) + binary
    masks, the 8 TEDM timesteps (train.py:47).  One step = B*8 UNet forwards (taps only, one native batch in '(b step)' order)
    + head forward/backward + Adam on the head; the head's 960 -> 128 projection runs fused at the taps' native resolution
    (tedm_amd/head.py), once per tap with the same weights for every timestep.  The unfused reference formulation
    (materialised (B, 960*8, S, S) features, regrouped to (B*8, 960, S, S)) and the inference call (`tedm.predict`: mean over
    timesteps of the sigmoids > .5, app.py:62-79) are timed next to it."""
    import torch
    from tedm_amd import DatasetDM, load_synth_weights, tedm
    steps = [1, 10, 25, 50, 200, 400, 600, 800]
    B, S = args.batch, 128
    ddm = DatasetDM(Namespace(normalize=True, dim=64, saved_diffusion_model='/nonexistent', verbose=False, device='cuda',
                              t_steps_to_save=steps, max_images_per_forward=256, shared_weights_over_timesteps=True))
    ddm.diffusion_model = dm
    load_synth_weights(ddm.classifier, seed=5)
    ddm = ddm.to(dev).train()
    dm.eval()
    T = len(steps)
    g = torch.Generator().manual_seed(99 + rank)
    x = torch.rand(B, 1, S, S, generator=g).to(dev)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing='ij')
    mask = ((((yy - 64) / 40.) ** 2 + ((xx - 40) / 22.) ** 2 < 1) | (((yy - 64) / 40.) ** 2 + ((xx - 88) / 22.) ** 2 < 1)).float()
    y = mask[None, None].expand(B, 1, S, S).contiguous().to(dev)                 # two blobs, like a lung mask
    y_rep = tedm.replicate_targets(y, T)
    head = list(ddm.classifier.parameters())
    opt = torch.optim.Adam(head, lr=1e-4)

    def step():
        opt.zero_grad()
        loss, _ = tedm.bce_loss(ddm(x), y_rep)
        loss.backward()
        tedm.allreduce_head_grads_(dist, head)
        opt.step()
        return loss.detach()
    res = {}
    for fused in (False, True):
        ddm.fused_head = fused
        torch.cuda.reset_peak_memory_stats(dev)
        elapsed, loss = dp.timed_steps(step, args.steps, args.warmup, dist, dev)
        res[fused] = (elapsed, float(loss), torch.cuda.max_memory_allocated(dev) / 2 ** 30)
    ddm.eval()
    with torch.no_grad():
        infer_s, (prob, m) = dp.timed_steps(lambda: tedm.predict(ddm, x), args.steps, args.warmup, dist, dev)
    dice = float(tedm.dice(m, y).nanmean())
    if rank == 0:
        el, loss, mem = res[True]
        print(json.dumps({
            'metric': 'TEDM head train-step images/sec @128x128, 8 timesteps (frozen UNet x8 + shared-weights head fwd/bwd + Adam)',
            'value': round(world * B * args.steps / el, 2), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * el / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.prec, 'data': 'synthetic',
            'config': {'workload': f'DatasetDM + shared-weights classifier (TEDM) train step, {B}x1x{S}x{S} per GPU, t_steps={steps} '
                                   f'(BASELINE configs[4] shape)', 'unet_forwards_per_step': B * T,
                       'final_loss': round(loss, 6), 'peak_mem_GiB': round(mem, 2),
                       'unfused_ms_per_step': round(1e3 * res[False][0] / args.steps, 3),
                       'unfused_peak_mem_GiB': round(res[False][2], 2),
                       'predict_ms_per_batch': round(1e3 * infer_s / args.steps, 3),
                       'predict_images_per_s': round(world * B * args.steps / infer_s, 2), 'dice_of_the_briefly_trained_head': round(dice, 4)}}),
              flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_sample(args, dm, dp, dist, dev, world, rank):
    """the serial ancestral-sampling chain (trainers/utils.py:62-98), `args.batch` images per GPU ("replicas only":
    the chain does not shard).  A step here is ONE denoise step x_t -> x_{t-1} (UNet forward + dynamic thresholding +
    posterior sample); the chain replays a hipGraph-captured step, the eager rate is reported next to it."""
    import torch
    B, S = args.batch, args.size
    n = max(args.steps, 1)
    x = torch.randn(B, 1, S, S, device=dev)
    res = {}
    with torch.no_grad():
        for use_graph in (False, True):
            dm.sample(B, S, steps=max(args.warmup, 2), use_graph=use_graph, x_T=x)           # warm-up (and capture cost for the graph)
            dp.barrier(dist, dev)
            t0 = time.perf_counter()
            out = dm.sample(B, S, steps=n, use_graph=use_graph, x_T=x)
            dp.barrier(dist, dev)
            res[use_graph] = dp.max_over_ranks(dist, time.perf_counter() - t0, dev)
    if rank == 0:
        el = res[True]
        print(json.dumps({
            'metric': 'sampler denoise steps/sec (one x_t -> x_{t-1} step of the ancestral chain, hipGraph replay)',
            'value': round(world * n / el, 2), 'unit': 'steps/s', 'n_gpus': world, 'steps': n, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * el / n, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.prec, 'data': 'synthetic',
            'config': {'workload': f'DiffusionModel.sample, {B}x1x{S}x{S} per GPU, {n} chain steps (includes one graph capture)',
                       'eager_ms_per_step': round(1e3 * res[False] / n, 3), 'images_per_s_full_chain_of_1000': round(world * B / (el / n * 1000), 3),
                       'finite': bool(torch.isfinite(out).all())}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--prec', default=os.environ.get('TEDM_PREC'), choices=['bf16x3', 'bf16', 'fp16x2', 'f16f4'],
                    help="arithmetic mode of the MFMA contractions; default: f16f4 (the library's opt-in fast mode; the module default is bf16x3)")
    ap.add_argument('--mode', default='denoise', choices=['denoise', 'extract', 'train', 'tedm', 'sample'],
                    help="'extract' = BASELINE config 3: DatasetDM.extract_features, 5 timesteps (extra mode; the default headline is 'denoise')")
    ap.add_argument('--no-cpu-baseline', '--no-extras', dest='no_cpu_baseline', action='store_true',
                    help='skip the legs outside the timed region (CPU baseline + eps check, two-stream pass): what the rocprofv3 runs use')
    ap.add_argument('--streams', type=int, default=1, help='inference: sub-batches on this many HIP streams (Unet.streams; bit-identical results)')
    ap.add_argument('--buckets', type=int, default=4, help='--mode train: gradient all-reduce buckets launched under the backward (1 = one all-reduce after it)')
    ap.add_argument('--breakdown', action='store_true', help='extra pass: HIP-event time of every kernel -> stderr')
    ap.add_argument('--tile-wide', type=int, default=None, help='tuning: conv tile id for Cout>=128')
    ap.add_argument('--tile-narrow', type=int, default=None, help='tuning: conv tile id for Cout<128')
    ap.add_argument('--tune', action='append', default=[], metavar='KEY=INT', help='tedm_set_tuning(KEY, INT)')
    args = ap.parse_args()

    import torch
    from tedm_amd import DiffusionModel, _lib, load_synth_weights

    from tedm_amd import dp
    world, rank, local = dp.env_world()
    if world != args.gpus:
        if rank == 0:
            print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run', file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    # TEDM_DIST_BACKEND=gloo + fewer GPUs than ranks = rehearsal of the multi-process path on a one-GPU box (ranks share the card)
    backend = os.environ.get('TEDM_DIST_BACKEND', 'nccl')
    dist = dp.init(backend) if world > 1 else None
    assert torch.cuda.is_available(), 'bench.py needs a ROCm GPU (the HIP path has no CPU fallback)'
    dev = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    L = _lib.lib()
    if args.tile_wide is not None:
        _lib.check(L.tedm_set_tuning(b'conv_tile_wide', args.tile_wide))
    if args.tile_narrow is not None:
        _lib.check(L.tedm_set_tuning(b'conv_tile_narrow', args.tile_narrow))

    for kv in args.tune:
        k, v = kv.split('=')
        _lib.check(L.tedm_set_tuning(k.encode(), int(v)), f'tune {kv}')

    B, S, K, W = args.batch, args.size, args.steps, args.warmup
    cfg = Namespace(normalize=True, dim=64, dim_mults=(1, 2, 4, 8), channels=1, timesteps=1000,
                    beta_schedule='cosine', objective='pred_noise')
    dm = DiffusionModel(cfg)
    load_synth_weights(dm.model, seed=0)
    dm = dm.to(dev).eval()
    if args.prec is None:          # the benchmark opts into the fast arithmetic f16f4 (inference: gated by the eps check below; training: forward
        args.prec = 'f16f4'        # and data gradients in f16f4, weight gradients bf16x3: gradients within 3e-3 per tensor / 5e-4 overall of
                                   # fp32 autograd, tests/test_gpu_train.py); `--prec bf16x3` is the library default (1e-4 per tensor)
    dm.model.precision = dm.model.train_precision = args.prec
    dm.model.streams = args.streams

    g = torch.Generator().manual_seed(1234 + rank)
    x0 = torch.rand(B, 1, S, S, generator=g).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g)
    t[0] = 0                       # the measured worst case of the eps error (profiles/*eps_margin.txt) is always in the checked batch
    if B > 1:
        t[1] = 999
    t = t.to(dev)
    noise = torch.randn(B, 1, S, S, generator=g).to(dev)

    def step():
        x_t = dm._q_sample(x0, t, noise, normalize=True)
        return dm.model(x_t, t)

    if args.mode == 'extract':
        return bench_extract(args, dm, dp, dist, dev, world, rank, x0, noise)
    if args.mode == 'train':
        return bench_train(args, dm, dp, dist, dev, world, rank, x0, t, noise)
    if args.mode == 'tedm':
        return bench_tedm(args, dm, dp, dist, dev, world, rank)
    if args.mode == 'sample':
        return bench_sample(args, dm, dp, dist, dev, world, rank)

    def sync_all():
        dp.barrier(dist, dev)

    with torch.no_grad():
        for _ in range(W):
            eps = step()
        # one more untimed step with HIP events around EVERY convolution launch: finds the dominant kernel symbol and the
        # convolutions' share of the step.  In the timed region only that one symbol is bracketed (an event pair costs ~5 us
        # of launch gap: around all 75 convolutions it would stretch the step by ~3 %).
        L.tedm_profile_reset(); L.tedm_profile_filter(None); L.tedm_profile_enable(1)
        step()
        torch.cuda.synchronize(dev)
        L.tedm_profile_enable(0)
        conv_all = sorted(_lib.profile_entries(), key=lambda e: -e[2])
        dominant = conv_all[0][0] if conv_all else ''
        # ---- timed region: exactly K steps, barrier + synchronize on both sides ----------------
        L.tedm_profile_reset()
        L.tedm_profile_filter(dominant.encode())
        L.tedm_profile_enable(1)                 # event pairs around the dominant kernel's launches only
        sync_all()
        t0 = time.perf_counter()
        for _ in range(K):
            eps = step()
        sync_all()
        elapsed = time.perf_counter() - t0
        L.tedm_profile_enable(0)
        L.tedm_profile_filter(None)
        conv_prof = _lib.profile_entries()
        elapsed = dp.max_over_ranks(dist, elapsed, dev)

        # outside the timed region: the same K steps with the batch cut into two sub-batches on two side HIP streams
        # (Unet.streams = 2, bit-identical results; kernel tails / launch gaps of one sub-batch are filled by the other).  Reported
        # NEXT to the headline number, which stays single-stream so that per-kernel durations are those of a kernel that owns the chip.
        multi = None
        if args.streams == 1 and args.mode == 'denoise' and B >= 2 * 8 and not args.no_cpu_baseline:   # --no-extras runs (profiling) time ONLY the K steps
            dm.model.streams = 2
            for _ in range(2):
                step()
            sync_all()
            m0 = time.perf_counter()
            for _ in range(K):
                step()
            sync_all()
            multi = dp.max_over_ranks(dist, time.perf_counter() - m0, dev)
            dm.model.streams = 1

        breakdown = None
        if args.breakdown and rank == 0:
            L.tedm_profile_reset()
            L.tedm_profile_enable(2)
            for _ in range(max(1, K // 4)):
                step()
            torch.cuda.synchronize(dev)
            L.tedm_profile_enable(0)
            breakdown = _lib.profile_entries()

    imgs = world * B * K
    value = imgs / elapsed
    ms_per_step = 1e3 * elapsed / K

    out = {
        'metric': 'denoise-step images/sec @256x256 bs32 per GPU (UNet eps forward)', 'value': round(value, 2),
        'unit': 'img/s', 'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(ms_per_step, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.prec,
        'data': 'synthetic',
        # what torch.distributed itself reports (N > 1: "nccl" IS RCCL on ROCm); the denoise path has no data-path collective,
        # only the barrier and the max-over-ranks of the elapsed time go through it
        'dist_backend': dist.get_backend() if dist is not None else None, 'world_size': dist.get_world_size() if dist is not None else 1,
        'config': {'workload': f'TEDM Unet(dim=64,dim_mults=(1,2,4,8),channels=1) q-sample + eps forward, '
                               f'{B}x1x{S}x{S} per GPU (BASELINE configs[1])',
                   'images_per_gpu': B, 'image_size': S, 'params': 36245377,
                   'precision': {'bf16x3': 'bf16 hi/lo split MFMA (3 products), fp32 accumulate, fp32 activations in HBM',
                                 'bf16': 'bf16 MFMA, fp32 accumulate, fp32 activations in HBM',
                                 'fp16x2': 'fp16 activations x fp16 hi/lo weights (2 products), fp32 accumulate, fp32 activations in HBM',
                                 'f16f4': 'fp16 x fp16 MFMA + one block-scaled fp4 MFMA (K=64) carrying both cross terms of the fp16 split: '
                                          '1.5 bf16-rate MFMA issues per 16 channels; the two 3x3 convolutions INSIDE every ResnetBlock below the full '
                                          'resolution (downs.1-3, mid, ups.0-2: 28 of the 46 k x k convolutions) run one fp16 product; '
                                          'fp32 accumulate, fp32 activations in HBM'}[args.prec],
                   'parallelism': f'dp{world} (batch sharded, no data-path collective)', 'hip_streams_per_gpu': args.streams},
    }

    if multi is not None:
        out['two_streams'] = {'value': round(world * B * K / multi, 2), 'unit': 'img/s', 'ms_per_step': round(1e3 * multi / K, 4),
                              'note': 'same K steps, batch as two sub-batches on two HIP streams (Unet.streams=2, bit-identical); outside the timed region'}
    if rank == 0:
        # ---- roofline of the dominant kernel, from HIP events recorded on the launch stream -------
        conv_prof.sort(key=lambda e: -e[2])
        if conv_prof:
            name, launches, ms, flops, byts = conv_prof[0]
            achieved = flops / (ms * 1e-3) / 1e12
            conv_ms = sum(e[2] for e in conv_all)                  # all convolutions: from the untimed profiled step
            step_ms_prof = 1e3 * elapsed / K
            # which roofline bounds this kernel: algorithmic flops at the dense MFMA peak vs algorithmic bytes at the HBM peak
            t_mfma, t_hbm = flops / (PEAK_BF16_TFLOPS * 1e12), byts / (PEAK_HBM_GBS * 1e9)
            hbm_bound = t_hbm > t_mfma
            gbs = byts / (ms * 1e-3) / 1e9
            def products_of(kernel_name):       # MFMA issues per algorithmic product: last template argument = TEDM_PREC_*
                import re
                m_ = re.search(r',(\d)>$', kernel_name)
                return {0: 3, 1: 1, 2: 2, 3: 1.5, 4: 1}[int(m_.group(1))] if m_ else {'bf16x3': 3, 'fp16x2': 2, 'bf16': 1, 'f16f4': 1.5}[args.prec]
            mp = products_of(name)
            out['roofline'] = {
                'bound': 'hbm' if hbm_bound else 'mfma', 'kernel': name,
                'achieved': round(gbs if hbm_bound else achieved, 2), 'peak': PEAK_HBM_GBS if hbm_bound else PEAK_BF16_TFLOPS,
                'unit': 'GB/s' if hbm_bound else 'TFLOP/s',
                'frac': round(gbs / PEAK_HBM_GBS if hbm_bound else achieved / PEAK_BF16_TFLOPS, 4), 'traffic': pmc_traffic_with_source(name)[0], 'traffic_source': pmc_traffic_with_source(name)[1],
                # `achieved`/`frac` of an HBM-bound kernel count the bytes of THIS build's storage (fp32 activations: 4 B per element
                # of SURVEY 8d's (Cin+Cout)*HW); the same elements at the survey's bf16 convention (2 B) and the MFMA-side fraction
                # are given next to it so that rounds stay comparable whatever the storage format
                'frac_fp32_storage_bytes': round(gbs / PEAK_HBM_GBS, 4), 'frac_survey_bf16_bytes': round(gbs / 2 / PEAK_HBM_GBS, 4),
                'frac_mfma_side': round(achieved / PEAK_BF16_TFLOPS, 4),
                'launches': launches, 'avg_us': round(1e3 * ms / launches, 2),
                'alg_gflop_per_launch': round(flops / launches / 1e9, 3),
                'alg_mb_per_launch': round(byts / launches / 1e6, 3),
                'alg_tflops': round(achieved, 2), 'alg_gbs': round(gbs, 1),
                'kernel_share_of_step': round(ms / (1e3 * elapsed), 4),
                'all_conv_share_of_step': round(conv_ms / step_ms_prof, 4),
                'all_conv_tflops': round(sum(e[3] for e in conv_all) / (conv_ms * 1e-3) / 1e12, 2),
                # `alg_tflops` counts ALGORITHMIC flops; per algorithmic product the arithmetic mode issues `mfma_products` bf16-rate
                # MFMAs (f16f4: two fp16 + one fp4 K=64 issue per 32 channels = 1.5 per 16), so the matrix pipes carry
                # `alg_tflops * mfma_products`; the chip sustains ~1.5 PF/s of fp16 MFMA issue under load (power-limited)
                'mfma_products': mp,
                'frac_of_issued_mfma_peak': round(achieved * mp / PEAK_BF16_TFLOPS, 4),
                # the convolution with the largest share among the MFMA-bound ones (Cout >= 128), same event data
                'mfma_bound_kernel': None,
            }
            for nm, ln, m_, fl, by in conv_all:
                if fl / (PEAK_BF16_TFLOPS * 1e12) >= by / (PEAK_HBM_GBS * 1e9):
                    out['roofline']['mfma_bound_kernel'] = {'kernel': nm, 'alg_tflops': round(fl / (m_ * 1e-3) / 1e12, 2),
                                                           'frac': round(fl / (m_ * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                                           'avg_us': round(1e3 * m_ / ln, 2), 'mfma_products': products_of(nm),
                                                           'from': 'untimed profiled step'}
                    break
        per_img = elapsed / (B * K)
        scale = (S / 256.0) ** 2
        out['whole_step'] = {
            't_roof_frac': round(T_ROOF_PER_IMG_256 * scale / per_img, 4),
            'mfma_frac': round(A_FLOPS_PER_IMG_256 * scale / per_img / (PEAK_BF16_TFLOPS * 1e12), 4),
            'hbm_frac_bf16_model': round(A_BYTES_PER_IMG_256 * scale / per_img / (PEAK_HBM_GBS * 1e9), 4),
            'us_per_image': round(per_img * 1e6, 2),
        }
        if breakdown:
            breakdown.sort(key=lambda e: -e[2])
            tot = sum(e[2] for e in breakdown)
            print('--- per-kernel HIP-event breakdown (separate pass) ---', file=sys.stderr)
            for name, n, ms, fl, by in breakdown:
                print(f'{ms / tot * 100:6.2f}%  {ms / n * 1e3:9.1f} us x {n:5d}  {fl / (ms * 1e-3) / 1e12 if ms else 0:8.1f} TF/s '
                      f'{by / (ms * 1e-3) / 1e9 if ms else 0:8.0f} GB/s  {name}', file=sys.stderr)

        # ---- CPU baseline (rank 0, N=1): the oracle restatement on the host cores, bounded sample; the same passes check eps
        # over EVERY image of the timed batch (image 0 is pinned to t = 0, image 1 to t = 999) ----
        if world == 1 and not args.no_cpu_baseline:
            import statistics
            from oracle import unet_ref as O
            ncores = host_cores()
            torch.set_num_threads(ncores)
            w = {k: v.detach().cpu() for k, v in dm.model.state_dict().items()}
            bufs = O.diffusion_buffers()
            nb = min(4, B)
            xc, tc, nc = x0.cpu(), t.cpu(), noise.cpu()
            eps_c = eps.cpu()
            rates, num, den, worst = [], 0.0, 0.0, (0.0, -1)
            with torch.no_grad():
                O.diffusion_forward(w, bufs, xc[:nb], tc[:nb], nc[:nb])          # 1 warm-up pass at the timed shape
                for b0 in range(0, B - nb + 1, nb):                               # B/nb timed passes (8 at B = 32): min & median
                    c0 = time.perf_counter()
                    ref = O.diffusion_forward(w, bufs, xc[b0:b0 + nb], tc[b0:b0 + nb], nc[b0:b0 + nb])
                    rates.append(nb / (time.perf_counter() - c0))
                    d = (eps_c[b0:b0 + nb].double() - ref.double())
                    num += float(d.pow(2).sum())
                    den += float(ref.double().pow(2).sum())
                    for i in range(nb):
                        e_i = float(d[i].norm() / ref[i].double().norm())
                        if e_i > worst[0]:
                            worst = (e_i, b0 + i)
                # BASELINE configs[0] shape (8 x 1 x 64 x 64, the reference's CPU-runnable plumbing case): 1 warm-up + 3 timed
                g1 = torch.Generator().manual_seed(4321)
                x1, t1, n1 = torch.rand(8, 1, 64, 64, generator=g1), torch.randint(0, 1000, (8,), generator=g1), torch.randn(8, 1, 64, 64, generator=g1)
                O.diffusion_forward(w, bufs, x1, t1, n1)
                r1 = []
                for _ in range(3):
                    c0 = time.perf_counter()
                    ref1 = O.diffusion_forward(w, bufs, x1, t1, n1)
                    r1.append(8 / (time.perf_counter() - c0))
                e1 = dm.model(dm._q_sample(x1.to(dev), t1.to(dev), n1.to(dev), normalize=True), t1.to(dev)).cpu()
                err1 = float((e1.double() - ref1.double()).norm() / ref1.double().norm())
            try:
                cpu_model = next(ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name'))
            except Exception:
                cpu_model = 'unknown'
            out['cpu_baseline'] = {'value': round(statistics.median(rates), 3), 'unit': 'img/s', 'cores': torch.get_num_threads(),
                                   'kind': 'port', 'cpu_model': cpu_model,
                                   'min': round(min(rates), 3), 'median': round(statistics.median(rates), 3), 'max': round(max(rates), 3),
                                   'timed_passes': len(rates),
                                   'sample': f'{len(rates)} timed passes of {nb}x1x{S}x{S} (all {len(rates) * nb} images of the GPU batch) after 1 '
                                             f'warm-up pass, fp32 q-sample+UNet forward (oracle/unet_ref.py, torch CPU ops); value = median',
                                   'config1_8x1x64x64': {'min': round(min(r1), 2), 'median': round(statistics.median(r1), 2), 'unit': 'img/s',
                                                         'timed_passes': 3, 'gpu_eps_rel_l2': float(f'{err1:.3e}')}}
            out['eps_rel_l2_vs_cpu'] = float(f'{(num / den) ** 0.5:.3e}')
            out['eps_check'] = {'images': len(rates) * nb, 'worst_sample_rel_l2': float(f'{worst[0]:.3e}'), 'worst_sample_index': worst[1],
                                'worst_sample_t': int(tc[worst[1]]), 't0_sample_included': bool((tc[:len(rates) * nb] == 0).any())}
            out['eps_gate'] = 1e-3
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
