#!/usr/bin/env python3
"""How much margin does the eps gate (1e-3 rel-L2 vs the CPU fp32 reference) have over inputs?  Per-sample eps error of the default
arithmetic on 256x256 images for several seeds at the extreme and at random timesteps: the SAME noised x_t (oracle q-sample) goes
through the HIP UNet and through the CPU oracle (test infrastructure: lives under tests/ because it imports the oracle).
usage (GPU box): python tests/eps_margin.py [n_batches] > profiles/<tag>_eps_margin.txt
                 python tests/eps_margin.py [n_batches] --trained [steps]     # the same on weights TRAINED here first:
                     the default UNet, `steps` (default 300) Adam steps of the build's own native training step on synthetic CXR-like
                     images (smooth random fields), then the margin scan, the per-module f16f4 errors of a 64x64 trace and a 30-step
                     sampling chain against the oracle chain on the trained weights."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from tedm_amd import Unet, load_synth_weights
from oracle import unet_ref as O


def cxr_like(B, S, g):
    """smooth random fields in [0, 1] (low-pass noise + a bright centre blob): the statistics of a chest film, not its content"""
    import torch.nn.functional as F
    lo = F.interpolate(torch.rand(B, 1, S // 16, S // 16, generator=g), size=(S, S), mode='bicubic', align_corners=False)
    mid = F.interpolate(torch.rand(B, 1, S // 4, S // 4, generator=g), size=(S, S), mode='bilinear', align_corners=False)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing='ij')
    blob = torch.exp(-(xx ** 2 / 0.5 + yy ** 2 / 0.8))[None, None]
    return (0.55 * lo + 0.2 * mid + 0.25 * blob).clamp(0, 1)


def train_weights(dev, steps):
    from argparse import Namespace
    from tedm_amd import DiffusionModel
    torch.manual_seed(0)
    dm = DiffusionModel(Namespace(normalize=True, dim=64, dim_mults=(1, 2, 4, 8), channels=1))     # torch's default initialisation,
    dm = dm.to(dev).train()                                                                       # as the reference's trainer starts from
    opt = torch.optim.Adam(dm.model.parameters(), lr=1e-4, fused=True)
    g = torch.Generator().manual_seed(2024)
    first = last = None
    for s in range(steps):
        x = cxr_like(16, 256, g).to(dev)
        opt.zero_grad(set_to_none=True)
        loss = dm.train_step_native(x)
        opt.step()
        if s == 0:
            first = float(loss)
        last = float(loss)
        if s % 50 == 0:
            print(f'  train step {s:4d} loss {float(loss):.4f}', flush=True)
    print(f'trained {steps} Adam steps (lr 1e-4, batch 16 x 256 x 256, synthetic CXR-like fields): loss {first:.4f} -> {last:.4f}')
    return dm


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    nb = int(args[0]) if args else 4
    trained = '--trained' in sys.argv
    dev = torch.device('cuda:0')
    prec = os.environ.get('TEDM_PREC', 'f16f4')                      # the opt-in fast mode is what needs the margin scan
    dm = None
    if trained:
        steps = int(args[1]) if len(args) > 1 else 300
        dm = train_weights(dev, steps)
        m = dm.model
        m.precision = prec
        m.eval()
    else:
        m = Unet(64, precision=prec)
        load_synth_weights(m, seed=0)
        m = m.to(dev).eval()
    w = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    bufs = O.diffusion_buffers()
    torch.set_num_threads(os.cpu_count() or 8)
    gr = torch.Generator().manual_seed(5)
    rep = m.fp16_range_report(O.q_sample(bufs, (cxr_like(2, 256, gr) if trained else torch.rand(2, 1, 256, 256, generator=gr)) * 2 - 1,
                                         torch.tensor([0, 999]), torch.randn(2, 1, 256, 256, generator=gr)).to(dev), torch.tensor([0, 999]).to(dev))
    print(f'fp16 range: max |activation| over the traced modules {max(rep.values()):.3g} at {max(rep, key=rep.get)} (f16f4 operands saturate at 65504)')
    allv = []
    print(f'precision {m.precision}; per-sample eps rel-L2 vs the CPU fp32 oracle, 256x256, '
          + ('weights TRAINED above' if trained else 'synthetic weights (seed 0)'))
    for s in range(nb):
        g = torch.Generator().manual_seed(100 + s)
        x0 = (cxr_like(4, 256, g) if trained else torch.rand(4, 1, 256, 256, generator=g)) * 2 - 1
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
    if not trained:
        return
    # per-module errors of the f16f4 trace on the trained weights (what tests/test_gpu_parity.py bounds with MODE_TOL['f16f4']['module'])
    g = torch.Generator().manual_seed(7)
    x = (cxr_like(2, 64, g) * 2 - 1)
    t = torch.tensor([20, 700])
    x_t = O.q_sample(bufs, x, t, torch.randn(2, 1, 64, 64, generator=g))
    with torch.no_grad():
        _, tr_o = O.unet_forward(w, x_t, t, return_trace=True)
        m.native_forward(x_t.to(dev), t.to(dev), trace=True)
        tr = m.last_trace
    worst = sorted(((float((tr[k].cpu().double() - tr_o[k].double()).norm() / tr_o[k].double().norm()), k) for k in tr_o if k in tr), reverse=True)
    print('per-module rel-L2 of the 64x64 trace, trained weights, worst five: ' + ', '.join(f'{k} {e:.2e}' for e, k in worst[:5]))
    # 30-step sampling chain from t = 999 (noise re-used on both sides), HIP f16f4 vs the oracle chain
    dm.eval()
    g = torch.Generator().manual_seed(11)
    xh = torch.randn(1, 1, 64, 64, generator=g)
    xo = xh.clone()
    worst_chain = 0.0
    with torch.no_grad():
        for i, tt in enumerate(range(999, 969, -1)):
            bt = torch.tensor([tt])
            noise = torch.randn(1, 1, 64, 64, generator=g)
            mean_h, logvar_h, _ = dm.p_mean_variance(xh.to(dev), bt.to(dev))
            xh = (mean_h + (0.5 * logvar_h).exp() * noise.to(dev)).cpu()
            mean_o, logvar_o, _ = O.p_mean_variance(w, bufs, xo, bt)
            xo = mean_o + (0.5 * logvar_o).exp() * noise
            worst_chain = max(worst_chain, float((xh.double() - xo.double()).norm() / xo.double().norm()))
    print(f'30-step sampling chain from t = 999 at 64x64 on the trained weights: worst x_t rel-L2 vs the oracle chain {worst_chain:.3e}')


if __name__ == '__main__':
    main()
