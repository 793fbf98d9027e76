"""ORACLE TOOLING -- runs ONLY in the build container (needs /root/reference); never shipped on the
product path and never executed on the GPU box.

Imports the reference implementation (mmr12/TEDM at /root/reference) with the two absent
third-party modules stubbed, loads the build's deterministic synthetic
weights into it, and writes small golden vectors (inputs + reference outputs) under tests/golden/.
It also checks the CPU restatement (oracle/unet_ref.py) against the reference on every case and
refuses to write goldens if the restatement deviates by more than 1e-5 rel-L2.

    python oracle/make_golden.py
"""
from __future__ import annotations

import json
import sys
import types
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
GOLD = ROOT / 'tests' / 'golden'


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, '/root/reference')
    tb = types.ModuleType('torch.utils.tensorboard')
    tb.SummaryWriter = type('SummaryWriter', (), {'__init__': lambda s, *a, **k: None})
    sys.modules['torch.utils.tensorboard'] = tb
    tv, tvu, tvt = (types.ModuleType(n) for n in ('torchvision', 'torchvision.utils', 'torchvision.transforms'))
    tvu.make_grid = None
    tvt.Resize = type('Resize', (), {'__init__': lambda s, *a, **k: None})
    tvt.InterpolationMode = type('InterpolationMode', (), {})
    tv.utils, tv.transforms = tvu, tvt
    sys.modules.update({'torchvision': tv, 'torchvision.utils': tvu, 'torchvision.transforms': tvt})
    from models.unet_model import Unet
    from models.diffusion_model import DiffusionModel
    from models.datasetDM_model import DatasetDM
    return Unet, DiffusionModel, DatasetDM


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def inputs(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.rand(B, 1, S, S, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn(B, 1, S, S, generator=g)
    return x0, t, noise


def cfg_ns(dim, **kw):
    return Namespace(normalize=True, dim=dim, dim_mults=(1, 2, 4, 8), channels=1, timesteps=1000,
                     beta_schedule='cosine', objective='pred_noise', p2_loss_weight_gamma=0., p2_loss_weight_k=1.,
                     saved_diffusion_model='/nonexistent', verbose=False, device='cpu', **kw)


def sub(t: torch.Tensor) -> np.ndarray:
    """deterministic subsample of an NCHW tap for compact fixtures: every 4th channel, every 2nd pixel"""
    return t[:, ::4, ::2, ::2].contiguous().numpy()


def main():
    from oracle import unet_ref as O
    from tedm_amd.weights import synth_state_dict
    RUnet, RDiff, RDM = import_reference()
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    GOLD.mkdir(parents=True, exist_ok=True)
    worst = 0.0

    # ---- 0. state-dict contract ---------------------------------------------------------------
    ref = RUnet(64, dim_mults=(1, 2, 4, 8), channels=1)
    keys = {k: list(v.shape) for k, v in ref.state_dict().items()}
    (GOLD / 'unet_d64_state_dict_keys.json').write_text(json.dumps(keys, indent=0))

    # ---- 1. default model: eps + decoder taps at 64x64 (B=2) and eps at 256x256 (B=1) ------------
    w64 = synth_state_dict({k: tuple(v) for k, v in keys.items()}, seed=0)
    ref.load_state_dict(w64)
    ref.eval()
    dm_ref = RDiff(cfg_ns(64))
    dm_ref.model.load_state_dict(w64)
    dm_ref.eval()
    bufs = O.diffusion_buffers()
    for name in bufs:
        assert torch.equal(bufs[name], getattr(dm_ref, name)), name
    feats = {}
    for i, lvl in enumerate(ref.ups):
        lvl[2].register_forward_hook(lambda m, inp, out, i=i: feats.__setitem__(i, out.detach().clone()))

    x0, t, noise = inputs(2, 64, 1234)
    x_t, _ = dm_ref.forward_diffusion_model(x0 * 2 - 1, t, noise)
    assert torch.equal(x_t, O.q_sample(bufs, x0 * 2 - 1, t, noise))
    eps_ref = ref(x_t, t)
    taps_ref = [feats[i] for i in range(4)]
    eps_o, taps_o = O.unet_forward(w64, x_t, t, return_taps=True)
    e = max([rel(eps_o, eps_ref)] + [rel(a, b) for a, b in zip(taps_o, taps_ref)])
    worst = max(worst, e)
    print(f'd64 s64: oracle vs reference rel-L2 {e:.3e}')
    np.savez(GOLD / 'eps_d64_s64.npz', x0=x0.numpy(), t=t.numpy(), noise=noise.numpy(), x_t=x_t.numpy(),
             eps=eps_ref.numpy(), **{f'tap{i}_sub': sub(taps_ref[i]) for i in range(4)},
             **{f'tap{i}_mean': taps_ref[i].mean(dim=(2, 3)).numpy() for i in range(4)})

    # timestep=None path (supervised baseline call, trainers/train_baseline.py:37)
    eps_nt = ref(x_t)
    e = rel(O.unet_forward(w64, x_t, None), eps_nt)
    worst = max(worst, e)
    print(f'd64 s64 t=None: oracle vs reference rel-L2 {e:.3e}')
    np.savez(GOLD / 'eps_d64_s64_notime.npz', x_t=x_t.numpy(), eps=eps_nt.numpy())

    # forward-only training loss (diffusion_model.py:120-143): noise reproduced from the torch seed
    torch.manual_seed(77)
    loss_ref = dm_ref.train_step(x0, t=t)
    torch.manual_seed(77)
    nz = torch.randn_like(x0)
    loss_o = O.train_loss(w64, bufs, x0, t, nz)
    assert abs(float(loss_o) - float(loss_ref)) < 1e-5, (float(loss_o), float(loss_ref))
    np.savez(GOLD / 'loss_d64_s64.npz', x0=x0.numpy(), t=t.numpy(), noise=nz.numpy(), loss=np.float32(loss_ref))

    x0, t, noise = inputs(1, 256, 4321)
    x_t = O.q_sample(bufs, x0 * 2 - 1, t, noise)
    eps_ref = ref(x_t, t)
    e = rel(O.unet_forward(w64, x_t, t), eps_ref)
    worst = max(worst, e)
    print(f'd64 s256: oracle vs reference rel-L2 {e:.3e}')
    np.savez(GOLD / 'eps_d64_s256.npz', x_t=x_t.numpy(), t=t.numpy(), eps=eps_ref.numpy())

    # ---- 2. mini model (dim 32) : full per-module trace at 16x16, B=2 --------------------------
    mini = RUnet(32, dim_mults=(1, 2, 4, 8), channels=1)
    kmini = {k: tuple(v.shape) for k, v in mini.state_dict().items()}
    w32 = synth_state_dict(kmini, seed=3)
    mini.load_state_dict(w32)
    mini.eval()
    trace_ref = {}
    named = {'init_conv': mini.init_conv, 'mid_block1': mini.mid_block1, 'mid_attn': mini.mid_attn,
             'mid_block2': mini.mid_block2, 'final_res_block': mini.final_res_block}
    for i in range(4):
        for j in range(4):
            named[f'downs.{i}.{j}'] = mini.downs[i][j]
            named[f'ups.{i}.{j}'] = mini.ups[i][j]
    for nm, mod in named.items():
        mod.register_forward_hook(lambda m, inp, out, nm=nm: trace_ref.__setitem__(nm, out.detach().clone()))
    x0, t, noise = inputs(2, 16, 99)
    x_t = O.q_sample(bufs, x0 * 2 - 1, t, noise)
    eps_ref = mini(x_t, t)
    eps_o, trace_o = O.unet_forward(w32, x_t, t, dim=32, return_trace=True)
    e = rel(eps_o, eps_ref)
    for nm, v in trace_ref.items():
        e = max(e, rel(trace_o[nm], v))
    worst = max(worst, e)
    print(f'd32 s16 trace ({len(trace_ref)} modules): oracle vs reference rel-L2 {e:.3e}')
    np.savez(GOLD / 'trace_d32_s16.npz', x_t=x_t.numpy(), t=t.numpy(), eps=eps_ref.numpy(),
             **{'m:' + nm: v.numpy() for nm, v in trace_ref.items()})

    # ---- 3. DatasetDM.extract_features (LEDM steps) on the mini model, 16x16, B=2 ---------------
    args = cfg_ns(32, t_steps_to_save=[50, 150, 250])
    dmm = RDM(args)
    dmm.diffusion_model.model.load_state_dict(w32)
    x0, _, noise = inputs(2, 16, 555)
    f_ref = dmm.extract_features(x0, noise=noise)
    f_o = O.extract_features(w32, bufs, x0, [50, 150, 250], noise, dim=32)
    e = rel(f_o, f_ref)
    worst = max(worst, e)
    print(f'd32 s16 extract_features {tuple(f_ref.shape)}: oracle vs reference rel-L2 {e:.3e}')
    np.savez(GOLD / 'extract_d32_s16.npz', x0=x0.numpy(), noise=noise.numpy(), steps=np.array([50, 150, 250]),
             feats_sub=f_ref[:, ::7].contiguous().numpy(), feats_mean=f_ref.mean(dim=(2, 3)).numpy())
    # ---- 3b. default-width DatasetDM (960*T channels): features + pixel-classifier logits, 16x16, B=2
    # (the classifier head is a "next" row of SURVEY.md 8f; the vector is kept for that work)
    args = cfg_ns(64, t_steps_to_save=[50, 150, 250])
    dm64 = RDM(args)
    dm64.diffusion_model.model.load_state_dict(w64)
    x0, _, noise = inputs(2, 16, 556)
    f_ref = dm64.extract_features(x0, noise=noise)
    f_o = O.extract_features(w64, bufs, x0, [50, 150, 250], noise)
    e = rel(f_o, f_ref)
    worst = max(worst, e)
    print(f'd64 s16 extract_features {tuple(f_ref.shape)}: oracle vs reference rel-L2 {e:.3e}')
    sd = dm64.classifier.state_dict()
    pc = synth_state_dict({k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point}, seed=5)
    sd.update(pc)
    dm64.classifier.load_state_dict(sd)
    dm64.eval()
    logits = dm64.classifier(f_ref)
    e = rel(O.pixel_classifier(sd, f_ref), logits)
    worst = max(worst, e)
    print(f'd64 s16 pixel classifier: oracle vs reference rel-L2 {e:.3e}')
    np.savez(GOLD / 'classifier_d64_s16.npz', x0=x0.numpy(), noise=noise.numpy(), steps=np.array([50, 150, 250]),
             feats_mean=f_ref.mean(dim=(2, 3)).numpy(), logits=logits.numpy())

    # ---- 4. three Adam steps of DiffusionModel.train_step (config 1: pins the native backward + Adam trace, tests/test_gpu_train.py)
    torch.set_grad_enabled(True)
    dm_tr = RDiff(cfg_ns(64))
    dm_tr.model.load_state_dict(w64)
    dm_tr.train()
    opt = torch.optim.Adam(dm_tr.parameters(), lr=1e-4, weight_decay=0)
    x0, _, _ = inputs(8, 64, 2024)
    tt = torch.randint(0, 1000, (3, 8), generator=torch.Generator().manual_seed(7))
    losses = []
    for s in range(3):
        opt.zero_grad()
        torch.manual_seed(1000 + s)
        loss = dm_tr.train_step(x0, t=tt[s])
        loss.backward()
        opt.step()
        losses.append(float(loss))
    torch.set_grad_enabled(False)
    csum = float(sum(p.double().sum() for p in dm_tr.model.parameters()))
    print('train trace', losses, csum)
    np.savez(GOLD / 'train3_d64_s64.npz', x0=x0.numpy(), t=tt.numpy(), noise_seeds=np.array([1000, 1001, 1002]),
             losses=np.array(losses, dtype=np.float64), param_sum=np.float64(csum))

    print(f'worst oracle-vs-reference rel-L2 over all cases: {worst:.3e}')
    if worst > 1e-5:
        raise SystemExit('oracle restatement deviates from the reference by more than 1e-5')


if __name__ == '__main__':
    main()
