#!/usr/bin/env python3
"""How much margin does the eps gate (1e-3 rel-L2 vs the CPU fp32 reference) have over inputs?  Per-sample eps error of the default
arithmetic on 256x256 images for several seeds at the extreme and at random timesteps: the SAME noised x_t (oracle q-sample) goes
through the HIP UNet and through the CPU oracle (test infrastructure: lives under tests/ because it imports the oracle).
usage (GPU box): python tests/eps_margin.py [n_batches] > profiles/<tag>_eps_margin.txt"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from tedm_amd import Unet, load_synth_weights
from oracle import unet_ref as O


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device('cuda:0')
    m = Unet(64, precision=os.environ.get('TEDM_PREC', 'f16f4'))     # the opt-in fast mode is what needs the margin scan
    load_synth_weights(m, seed=0)
    m = m.to(dev).eval()
    w = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    bufs = O.diffusion_buffers()
    torch.set_num_threads(os.cpu_count() or 8)
    allv = []
    print(f'precision {m.precision}; per-sample eps rel-L2 vs the CPU fp32 oracle, 256x256, synthetic weights (seed 0)')
    for s in range(nb):
        g = torch.Generator().manual_seed(100 + s)
        x0 = torch.rand(4, 1, 256, 256, generator=g) * 2 - 1
        t = torch.tensor([0, 999, int(torch.randint(1, 999, (1,), generator=g)), int(torch.randint(1, 999, (1,), generator=g))])
        x_t = O.q_sample(bufs, x0, t, torch.randn(4, 1, 256, 256, generator=g))
        with torch.no_grad():
            eps = m(x_t.to(dev), t.to(dev)).cpu()
            ref = O.unet_forward(w, x_t, t)
        for i in range(4):
            e = float((eps[i].double() - ref[i].double()).norm() / ref[i].double().norm())
            allv.append(e)
            print(f'seed {100 + s} sample {i} t={int(t[i]):4d}  eps rel-L2 {e:.3e}', flush=True)
    allv.sort()
    print(f'n={len(allv)} median {allv[len(allv) // 2]:.3e} max {allv[-1]:.3e} (gate 1e-3)')


if __name__ == '__main__':
    main()
