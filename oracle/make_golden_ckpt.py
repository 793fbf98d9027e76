"""ORACLE TOOLING (build container only, needs /root/reference): a REFERENCE-WRITTEN checkpoint the HIP kernels accept.

SURVEY.md 8 f4 / VERDICT r3 item 2: the round-3 fixture (`reference_checkpoint_d8.pt`, dim = 8) pins the loader on the CPU but
cannot run on the GPU (the kernels need level widths that are multiples of 32).  This one is `dim = 32, dim_mults = (1, 2)`:
the imported reference `DiffusionModel` after two Adam steps (trainers/train_CXR14.py:28-32), written by the reference's own
`save` (trainers/train_base_diffusion.py:164-170; same dict as train_CXR14.py:96-102: Namespace + PosixPath inside), plus
  * the eps the reference computes from it on a 64x64 probe (models/diffusion_model.py:174 through models/unet_model.py:333-368);
  * what the reference `DatasetDM(args.saved_diffusion_model = <that file>)` (models/datasetDM_model.py:38-43) extracts from
    the probe images at two timesteps (:67-83), subsampled.
To keep the fixture at 3.3 MB the Adam moments are NOT stored: `save` is handed a fresh optimizer over the trained parameters
(the populated optimizer state of a reference checkpoint is what the d8 fixture covers).

    python oracle/make_golden_ckpt.py            # the checkpoint fixtures
    python oracle/make_golden_ckpt.py --learned  # learned_sinusoidal_cond golden (models/unet_model.py:96-114)
"""
import sys
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
from make_golden import GOLD, import_reference, inputs, rel   # noqa: E402

STEPS = [50, 400]


def main():
    from oracle import unet_ref as O
    RUnet, RDiff, RDM = import_reference()
    from trainers.train_base_diffusion import save as ref_save
    torch.manual_seed(11)
    cfg = Namespace(normalize=True, dim=32, dim_mults=(1, 2), channels=1, timesteps=1000, beta_schedule='cosine',
                    objective='pred_noise', p2_loss_weight_gamma=0., p2_loss_weight_k=1., img_size=64, lr=1e-4,
                    weight_decay=0.0, device='cpu', log_dir=Path('logs/tedm_fixture_d32'), experiment='img_only', seed=11)
    dmr = RDiff(cfg)
    opt = torch.optim.Adam(dmr.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)
    xb, _, _ = inputs(4, 32, 41)
    for s in range(2):
        opt.zero_grad()
        torch.manual_seed(300 + s)
        loss = dmr.train_step(xb)
        loss.backward()
        opt.step()
    path = GOLD / 'reference_checkpoint_d32.pt'
    ref_save(dmr, torch.optim.Adam(dmr.parameters(), lr=cfg.lr), cfg, path, 2)        # trainers/train_base_diffusion.py:164-170
    torch.set_grad_enabled(False)
    dmr.eval()
    xq, tq, nq = inputs(2, 64, 42)
    x_t, _ = dmr.forward_diffusion_model(xq * 2 - 1, tq, nq)
    eps = dmr.model(x_t, tq)
    w = {k: v.detach() for k, v in dmr.model.state_dict().items()}
    e = rel(O.unet_forward(w, x_t, tq, dim=32, dim_mults=(1, 2)), eps)

    # the reference DatasetDM loading that very file (models/datasetDM_model.py:38-43) and extracting features (:67-83)
    args = Namespace(**{**vars(cfg), 'saved_diffusion_model': str(path), 'verbose': False, 't_steps_to_save': STEPS})
    import argparse, pathlib
    # (the reference calls torch.load(path, map_location) as written for torch 1.13, where weights_only defaulted to False;
    #  under torch >= 2.6 the two non-tensor types inside are allow-listed instead)
    with torch.serialization.safe_globals([argparse.Namespace, pathlib.PosixPath, pathlib.PurePosixPath, pathlib.Path]):
        rdm = RDM(args)
    feats = rdm.extract_features(xq, noise=nq)                                          # (2, 96 * 2, 64, 64)
    np.savez(GOLD / 'reference_checkpoint_d32_probe.npz', x0=xq.numpy(), noise=nq.numpy(), x_t=x_t.numpy(), t=tq.numpy(),
             eps=eps.numpy(), steps=np.array(STEPS), feats_sub=feats[:, ::3, ::2, ::2].contiguous().numpy(),
             feats_mean=feats.mean(dim=(2, 3)).numpy(), feats_shape=np.array(feats.shape),
             n_params=np.int64(sum(p.numel() for p in dmr.parameters())),
             param_sum=np.float64(sum(float(p.double().sum()) for p in dmr.parameters())))
    print(f'reference checkpoint: {path.stat().st_size} bytes, features {tuple(feats.shape)}, eps oracle vs reference rel-L2 {e:.3e}')
    if e > 1e-5:
        raise SystemExit('oracle restatement deviates from the reference by more than 1e-5')


if __name__ == '__main__' and '--learned' not in sys.argv:
    main()


def learned_sinusoid_golden():
    """`learned_sinusoidal_cond=True` (models/unet_model.py:96-114,278-285) -- the reference Unet with the
    build's synthetic weights (incl. the learned frequency vector), eps on a 32x32 probe and the gradient of the sum-of-eps * R
    probe w.r.t. `time_mlp.0.weights`; the oracle restatement must agree."""
    from oracle import unet_ref as O
    from tedm_amd.weights import synth_state_dict
    RUnet, _, _ = import_reference()
    torch.manual_seed(0)
    ref = RUnet(32, dim_mults=(1, 2), channels=1, learned_sinusoidal_cond=True, learned_sinusoidal_dim=16)
    keys = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    w = synth_state_dict(keys, seed=3)
    ref.load_state_dict(w)
    x0, t, R = inputs(2, 32, 77)
    x = x0 * 2 - 1
    with torch.enable_grad():
        eps = ref(x, t)
        (eps * R).sum().backward()
    gw = ref.time_mlp[0].weights.grad.clone()
    g1 = ref.time_mlp[1].weight.grad.clone()
    with torch.no_grad():
        e = rel(O.unet_forward({k: v.detach() for k, v in ref.state_dict().items()}, x, t, dim=32, dim_mults=(1, 2)), eps.detach())
    np.savez(GOLD / 'learned_sinusoid_d32_s32.npz', x=x.numpy(), t=t.numpy(), R=R.numpy(), eps=eps.detach().numpy(),
             grad_weights=gw.numpy(), grad_mlp1=g1.numpy(), keys=np.array(sorted(keys)), fourier_in=np.int64(ref.time_mlp[1].in_features))
    print(f'learned sinusoid: {len(keys)} keys, time_mlp.1 in_features {ref.time_mlp[1].in_features}, eps oracle vs reference rel-L2 {e:.3e}')
    if e > 1e-5:
        raise SystemExit('oracle restatement deviates from the reference by more than 1e-5')


if __name__ == '__main__' and '--learned' in sys.argv:
    learned_sinusoid_golden()
