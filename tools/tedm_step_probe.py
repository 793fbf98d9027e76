#!/usr/bin/env python3
"""Diagnostic (GPU): kernel table of ONE TEDM train step (bench.py --mode tedm, fused head only) from torch.profiler.
usage: python tools/tedm_step_probe.py [batch]"""
import os, sys
from argparse import Namespace
import torch
sys.path.insert(0, os.getcwd())
import tedm_amd
from tedm_amd import DatasetDM, DiffusionModel, load_synth_weights, tedm

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S, steps = 128, [1, 10, 25, 50, 200, 400, 600, 800]
dev = torch.device('cuda:0')
cfg = Namespace(normalize=True, dim=64, dim_mults=(1, 2, 4, 8), channels=1, timesteps=1000, beta_schedule='cosine', objective='pred_noise')
dm = DiffusionModel(cfg)
load_synth_weights(dm.model, seed=0)
dm = dm.to(dev).eval()
dm.model.precision = dm.model.train_precision = 'f16f4'
ddm = DatasetDM(Namespace(normalize=True, dim=64, saved_diffusion_model='/nonexistent', verbose=False, device='cuda',
                          t_steps_to_save=steps, max_images_per_forward=256, shared_weights_over_timesteps=True))
ddm.diffusion_model = dm
load_synth_weights(ddm.classifier, seed=5)
ddm = ddm.to(dev).train()
dm.eval()
x = torch.rand(B, 1, S, S, device=dev)
y = (torch.rand(B, 1, S, S, device=dev) > 0.5).float()
y_rep = tedm.replicate_targets(y, len(steps))
head = list(ddm.classifier.parameters())
opt = torch.optim.Adam(head, lr=1e-4)
ddm.fused_head = True


def step():
    opt.zero_grad()
    loss, _ = tedm.bce_loss(ddm(x), y_rep)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=40, max_name_column_width=70))
