#!/usr/bin/env python3
"""BASELINE config 5 in miniature, end to end on ONE or N GPUs: TEDM = frozen diffusion UNet + pixel-classifier head
(trainers/train_datasetDM.py / train_baseline.py semantics) on JSRT-shaped SYNTHETIC data, Dice on held-out images.

    python examples/tedm_synthetic.py [--steps 60] [--size 128] [--train 32] [--val 8] [--batch 8]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/tedm_synthetic.py ...

Data: chest-like images = smooth background + two brighter "lung" ellipses of random position/size + noise; target = the
ellipse mask (what `dataloaders/JSRT.py` yields as (image, mask), both in [0,1]).  Model: `tedm_amd.DatasetDM` with a
randomly initialised frozen `DiffusionModel` (no trained checkpoint ships with the reference), the 8 TEDM timesteps
(train.py:47) and the SHARED-WEIGHTS classifier (trainers/train_datasetDM.py:30-42); only `classifier` is trained, through
`tedm_amd.tedm.train_head` (= trainers/train_baseline.py:17-96: replicated targets, BCE-with-logits, Adam lr 1e-4 (config.py));
prediction = mean over the timesteps of the sigmoids > 0.5 (app.py:79), Dice / precision / recall as train_baseline.py:146-161.
Under torch.distributed every rank trains on its shard, the head's gradients cross ranks in ONE flat all-reduce (BatchNorm
statistics stay per rank)."""
import argparse
import os
import sys
from argparse import Namespace

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def make_data(n, S, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing='ij')
    xs, ys = [], []
    for _ in range(n):
        cy = S * (0.5 + 0.08 * (torch.rand(1, generator=g) - 0.5))
        ry, rx = S * (0.28 + 0.06 * torch.rand(1, generator=g)), S * (0.14 + 0.04 * torch.rand(1, generator=g))
        c1, c2 = S * (0.31 + 0.04 * torch.rand(1, generator=g)), S * (0.69 - 0.04 * torch.rand(1, generator=g))
        mask = ((((yy - cy) / ry) ** 2 + ((xx - c1) / rx) ** 2 < 1) | (((yy - cy) / ry) ** 2 + ((xx - c2) / rx) ** 2 < 1)).float()
        bg = 0.35 + 0.15 * torch.sin(xx / S * 3.1) * torch.cos(yy / S * 2.3)
        img = (bg + 0.3 * mask + 0.05 * torch.randn(S, S, generator=g)).clamp(0, 1)
        xs.append(img[None])
        ys.append(mask[None])
    return torch.stack(xs), torch.stack(ys)


def run(steps=60, size=128, n_train=32, n_val=8, batch=8, lr=1e-4, seed=0, verbose=True, shared=True):
    """Returns the list of validation dicts (tedm.validate_head: val/loss, val/dice, val/precision, val/recall, val/dice_ensemble),
    first entry = before training."""
    from tedm_amd import DatasetDM, dp, tedm
    world, rank, local = dp.env_world()
    dist = dp.init(os.environ.get('TEDM_DIST_BACKEND', 'nccl')) if world > 1 else None
    dev = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    torch.manual_seed(seed)                                   # same head initialisation on every rank
    cfg = Namespace(normalize=True, dim=64, saved_diffusion_model='/nonexistent', verbose=False, device=str(dev),
                    t_steps_to_save=[1, 10, 25, 50, 200, 400, 600, 800],          # train.py:47
                    shared_weights_over_timesteps=shared, experiment='TEDM', lr=lr, weight_decay=0.0,
                    log_freq=max(1, steps // 4), val_freq=max(1, steps // 4), max_steps=steps, max_val_steps=-1, debug=False,
                    early_stop=False, log_dir=os.path.join(os.environ.get('TMPDIR', '/tmp'), f'tedm_synth_rank{rank}'))
    model = DatasetDM(cfg).to(dev)
    for p in model.diffusion_model.parameters():
        p.requires_grad_(False)
    opt = torch.optim.Adam(model.classifier.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)   # train_datasetDM.py:46
    xtr, ytr = make_data(n_train, size, 100)
    xva, yva = make_data(n_val, size, 200)
    beg, end = dp.shard_range(n_train, world, rank)
    xtr, ytr = xtr[beg:end], ytr[beg:end]
    g = torch.Generator().manual_seed(seed + rank)

    class Loader:                                             # a shuffling (x, y) loader, like dataloaders/JSRT.py's DataLoader
        def __iter__(self):
            perm = torch.randperm(xtr.shape[0], generator=g)
            for i in range(0, len(perm) - batch + 1, batch) or [0]:
                idx = perm[i:i + batch]
                yield xtr[idx], ytr[idx]
    val = [(xva[i:i + batch], yva[i:i + batch]) for i in range(0, n_val, batch)]
    hist = [tedm.validate_head(cfg, model, val, dist)]

    class Logger:
        def log(self, d, step):
            if 'val/loss' in d:
                hist.append(d)
                if verbose and rank == 0:
                    print(f"step {step:4d}  val loss {d['val/loss']:.4f}  dice/step {d['val/dice']:.3f}  precision {d['val/precision']:.3f}  "
                          f"recall {d['val/recall']:.3f}  dice(ensemble) {d['val/dice_ensemble']:.3f}", flush=True)
    tedm.train_head(cfg, model, opt, Loader(), val, Logger(), dist=dist)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return hist


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--train', type=int, default=32)
    ap.add_argument('--val', type=int, default=8)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--ledme', action='store_true', help='the LEDMe head (one classifier on the 960*8-channel concat) instead of TEDM')
    a = ap.parse_args()
    h = run(a.steps, a.size, a.train, a.val, a.batch, a.lr, shared=not a.ledme)
    print(f"held-out Dice (mean-over-steps mask): {h[0]['val/dice_ensemble']:.3f} -> {h[-1]['val/dice_ensemble']:.3f}")
