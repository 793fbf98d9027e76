#!/usr/bin/env python3
"""BASELINE configs 1 / 4 in miniature: the reference diffusion training loop (trainers/train_CXR14.py:16-60) over the native
train step, on synthetic CXR-like images, 1..N GPUs (one process per GPU, batch sharded, one flat-gradient all-reduce).

    python examples/train_diffusion_synthetic.py [--steps 20] [--size 64] [--batch 8]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/train_diffusion_synthetic.py ...
"""
import argparse
import os
import sys
import tempfile
from argparse import Namespace

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU')
    ap.add_argument('--dim', type=int, default=64)
    a = ap.parse_args()
    from tedm_amd import DiffusionModel, dp, trainer
    world, rank, local = dp.env_world()
    dist = dp.init(os.environ.get('TEDM_DIST_BACKEND', 'nccl')) if world > 1 else None
    dev = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    cfg = Namespace(normalize=True, dim=a.dim, dim_mults=(1, 2, 4, 8), channels=1, timesteps=1000, beta_schedule='cosine',
                    objective='pred_noise', device=str(dev), lr=1e-4, log_freq=5, val_freq=10, max_val_steps=2, max_steps=a.steps,
                    debug=False, log_dir=tempfile.mkdtemp(prefix='tedm_run_'), resume_path=None)
    torch.manual_seed(0)                                            # identical replicas
    model = DiffusionModel(cfg).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, fused=True)
    g = torch.Generator().manual_seed(10 + rank)
    low = torch.rand(4 * a.batch, 1, a.size // 8, a.size // 8, generator=g)
    imgs = torch.nn.functional.interpolate(low, size=(a.size, a.size), mode='bilinear')     # smooth "CXR-like" images in [0,1]
    loader = [imgs[i:i + a.batch] for i in range(0, imgs.shape[0], a.batch)]
    trainer.train(cfg, model, opt, loader, loader[:2], logger=None, step=0, dist=dist)
    if rank == 0:
        print('checkpoint:', os.path.join(cfg.log_dir, 'best_model.pt'), os.path.exists(os.path.join(cfg.log_dir, 'best_model.pt')))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
