#!/usr/bin/env python3
"""BASELINE config 5 in miniature, end to end on ONE or N GPUs: TEDM = frozen diffusion UNet + pixel-classifier head
(trainers/train_datasetDM.py / train_baseline.py semantics) on JSRT-shaped SYNTHETIC data, Dice on held-out images.

    python examples/tedm_synthetic.py [--steps 60] [--size 128] [--train 32] [--val 8] [--batch 8]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/tedm_synthetic.py ...

Data: chest-like images = smooth background + two brighter "lung" ellipses of random position/size + noise; target = the
ellipse mask (what `dataloaders/JSRT.py` yields as (image, mask), both in [0,1]).  Model: `tedm_amd.DatasetDM` with a
randomly initialised frozen `DiffusionModel` (no trained checkpoint ships with the reference) and the 8 TEDM timesteps
(train.py:47); only `classifier` is trained: BCE-with-logits (train_baseline.py:44), Adam lr 1e-4 (config.py), prediction
`sigmoid > 0.5`, Dice / precision / recall as train_baseline.py:146-161.  Under torch.distributed every rank trains on its
shard, classifier gradients are averaged with one all-reduce per parameter (BatchNorm statistics stay per rank)."""
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


def dice_pr(y_hat, y):
    inter = (y_hat & y).flatten(1).sum(1).float()
    d = 2 * inter / (y_hat.flatten(1).sum(1) + y.flatten(1).sum(1)).float()
    return d.nanmean().item(), (inter / y_hat.flatten(1).sum(1).float()).nanmean().item(), (inter / y.flatten(1).sum(1).float()).nanmean().item()


def run(steps=60, size=128, n_train=32, n_val=8, batch=8, lr=1e-4, seed=0, verbose=True):
    from tedm_amd import DatasetDM, dp
    world, rank, local = dp.env_world()
    dist = dp.init(os.environ.get('TEDM_DIST_BACKEND', 'nccl')) if world > 1 else None
    dev = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    torch.manual_seed(seed)                                   # same head initialisation on every rank
    model = DatasetDM(Namespace(normalize=True, dim=64, saved_diffusion_model='/nonexistent', verbose=False, device='cuda',
                                t_steps_to_save=[1, 10, 25, 50, 200, 400, 600, 800])).to(dev)
    model.diffusion_model.eval()
    for p in model.diffusion_model.parameters():
        p.requires_grad_(False)
    opt = torch.optim.Adam(model.classifier.parameters(), lr=lr)
    xtr, ytr = make_data(n_train, size, 100)
    xva, yva = make_data(n_val, size, 200)
    beg, end = dp.shard_range(n_train, world, rank)
    xtr, ytr = xtr[beg:end].to(dev), ytr[beg:end].to(dev)
    xva, yva = xva.to(dev), yva.to(dev)

    def validate():
        model.eval()
        with torch.no_grad():
            pred = torch.cat([model(xva[i:i + batch]) for i in range(0, n_val, batch)])
        model.train()
        model.diffusion_model.eval()
        d, p, r = dice_pr(torch.sigmoid(pred) > 0.5, yva > 0.5)
        return float(F.binary_cross_entropy_with_logits(pred, yva)), d, p, r

    model.train()
    model.diffusion_model.eval()
    hist = [validate()]
    g = torch.Generator().manual_seed(seed + rank)
    for step in range(1, steps + 1):
        idx = torch.randint(0, xtr.shape[0], (min(batch, xtr.shape[0]),), generator=g).to(dev)
        opt.zero_grad()
        loss = F.binary_cross_entropy_with_logits(model(xtr[idx]), ytr[idx])
        loss.backward()
        if dist is not None:
            for p in model.classifier.parameters():
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                p.grad.div_(world)
        opt.step()
        if step % max(1, steps // 4) == 0 or step == steps:
            hist.append(validate())
            if verbose and rank == 0:
                print(f'step {step:4d}  train loss {float(loss.detach()):.4f}  val loss {hist[-1][0]:.4f}  dice {hist[-1][1]:.3f}  '
                      f'precision {hist[-1][2]:.3f}  recall {hist[-1][3]:.3f}', flush=True)
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
    a = ap.parse_args()
    h = run(a.steps, a.size, a.train, a.val, a.batch, a.lr)
    print(f'held-out Dice: {h[0][1]:.3f} -> {h[-1][1]:.3f}')
