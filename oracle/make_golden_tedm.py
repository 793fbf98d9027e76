"""ORACLE TOOLING (build container only, needs /root/reference): round-3 golden vectors.

1. TEDM proper: the imported reference `DatasetDM` with the SHARED-WEIGHTS classifier swapped in exactly
   as trainers/train_datasetDM.py:30-42 does, at (2,1,64,64), T = 3: eval-mode logits (B*T,1,S,S), the mean-over-steps mask
   (auxiliary/postprocessing/testing_shared_weights.py:113-133, app.py:79), the replicated-target BCE loss
   (trainers/train_baseline.py:30,44-45), Dice / precision / recall of the reference's own functions (:146-161), and one
   train-mode forward/backward (BatchNorm on batch statistics) with the classifier gradients.
2. BASELINE configs[2] at fixture size: `extract_features` with the 5 default timesteps (config.py:78) at 16x16.
3. A REFERENCE-WRITTEN checkpoint (trainers/train_base_diffusion.py:164-170 `save`, the function trainers/train_baseline
   imports; same dict as train_CXR14.py:96-102): a tiny DiffusionModel after two Adam steps, Namespace + PosixPath inside,
   plus the eps the reference computes from it -- pins checkpoint.load_checkpoint / trainer.load.

Every case is also run through oracle/unet_ref.py and must agree to 1e-5 rel-L2.

    python oracle/make_golden_tedm.py
"""
import sys
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
from make_golden import GOLD, cfg_ns, import_reference, inputs, rel   # noqa: E402

TEDM_STEPS = [50, 150, 250]


def lung_like_masks(B, S, seed):
    """binary (B,1,S,S) float masks: two ellipses per image (what dataloaders/JSRT.py yields as y)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing='ij')
    out = []
    for _ in range(B):
        cy = S * (0.5 + 0.1 * (torch.rand(1, generator=g) - 0.5))
        ry, rx = S * (0.25 + 0.08 * torch.rand(1, generator=g)), S * (0.13 + 0.05 * torch.rand(1, generator=g))
        c1, c2 = S * 0.32, S * 0.68
        out.append(((((yy - cy) / ry) ** 2 + ((xx - c1) / rx) ** 2 < 1) | (((yy - cy) / ry) ** 2 + ((xx - c2) / rx) ** 2 < 1)).float()[None])
    return torch.stack(out)


def main():
    from oracle import unet_ref as O
    from tedm_amd.weights import synth_state_dict
    RUnet, RDiff, RDM = import_reference()
    import torch.nn as nn
    from einops import rearrange
    from einops.layers.torch import Rearrange
    from trainers import train_baseline as TB
    from trainers.train_base_diffusion import save as ref_save
    torch.manual_seed(0)
    bufs = O.diffusion_buffers()
    worst = 0.0

    # ---- 1. TEDM shared-weights head ------------------------------------------------------------------------
    torch.set_grad_enabled(False)
    args = cfg_ns(64, t_steps_to_save=TEDM_STEPS)
    model = RDM(args)
    w64 = synth_state_dict({k: tuple(v.shape) for k, v in model.diffusion_model.model.state_dict().items()}, seed=0)
    model.diffusion_model.model.load_state_dict(w64)
    model.classifier = nn.Sequential(                                     # trainers/train_datasetDM.py:33-42, verbatim structure
        Rearrange('b (step act) h w -> (b step) act h w', step=len(model.steps)),
        nn.Conv2d(960, 128, 1), nn.ReLU(), nn.BatchNorm2d(128), nn.Conv2d(128, 32, 1), nn.ReLU(), nn.BatchNorm2d(32),
        nn.Conv2d(32, 1, 1))
    sd = model.classifier.state_dict()
    sd.update(synth_state_dict({k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point}, seed=6))
    g = torch.Generator().manual_seed(61)                                 # non-trivial BatchNorm running statistics
    for k in ('3', '6'):
        sd[f'{k}.running_mean'] = 0.2 * torch.randn(sd[f'{k}.running_mean'].shape, generator=g)
        sd[f'{k}.running_var'] = 0.5 + torch.rand(sd[f'{k}.running_var'].shape, generator=g)
    model.classifier.load_state_dict(sd)
    model.eval()
    x0, _, noise = inputs(2, 64, 4242)
    y = lung_like_masks(2, 64, 7)
    T = len(TEDM_STEPS)
    feats = model.extract_features(x0, noise=noise)                       # (2, 2880, 64, 64)
    # centre the random head's logits (output bias := -median) so that the thresholded masks / Dice are not degenerate
    sd['7.bias'] = sd['7.bias'] - model.classifier(feats).median()
    model.classifier.load_state_dict(sd)
    bn = {k: v.clone().numpy().copy() for k, v in sd.items() if 'running' in k or k == '7.bias'}   # snapshot: the train-mode pass below updates the statistics
    logits = model.classifier(feats)                                      # (6, 1, 64, 64), rows (b step)
    assert tuple(logits.shape) == (2 * T, 1, 64, 64)
    f_o = O.extract_features(w64, bufs, x0, TEDM_STEPS, noise)
    pc = {k: v for k, v in sd.items() if v.dtype.is_floating_point}
    lo = O.tedm_classifier(pc, f_o, T)
    e = rel(lo, logits)
    worst = max(worst, e)
    print(f'TEDM eval logits: oracle vs reference rel-L2 {e:.3e}')
    # ensembling: testing_shared_weights.py:113-133
    y_hats = rearrange(torch.sigmoid(logits), '(b step) 1 h w -> step b 1 h w', step=T)
    prob = y_hats.mean(0)
    mask = prob > .5
    po, mo = O.tedm_ensemble(lo, T)
    assert torch.equal(mo, mask) and rel(po, prob) < 1e-6
    # app.py:79 on ONE image: sigmoid(TEDM(img)).mean(0) > .5
    one = torch.sigmoid(model.classifier(model.extract_features(x0[:1], noise=noise[:1]))).mean(0) > .5
    assert torch.equal(one, mask[0]), 'app.py single-image ensembling == batched regrouping'
    # metrics of the reference's own functions, ensembled and per timestep
    met = {'dice': TB.dice(mask, y), 'precision': TB.precision(mask, y), 'recall': TB.recall(mask, y)}
    met_t = {f'dice_t{i}': TB.dice(y_hats[i] > .5, y) for i in range(T)}
    assert torch.allclose(O.dice(mask, y), met['dice'], equal_nan=True) and torch.allclose(O.precision(mask, y), met['precision'], equal_nan=True) \
        and torch.allclose(O.recall(mask, y), met['recall'], equal_nan=True)
    # replicated-target loss, eval mode (validate(): trainers/train_baseline.py:121,125)
    from einops import repeat, reduce
    y_rep = repeat(y, 'b c h w -> (b step) c h w', step=T)
    assert torch.equal(y_rep, O.replicate_targets(y, T))
    exp_loss = reduce(torch.nn.functional.binary_cross_entropy_with_logits(logits, y_rep, reduction='none'), 'b c h w -> b c', 'mean')
    loss_eval = exp_loss.mean()
    loss_per_t = reduce(exp_loss, '(b step) c -> step', 'mean', step=T)                   # train_baseline.py:57
    lo_loss, lo_exp = O.bce_loss(lo, O.replicate_targets(y, T))
    assert abs(float(lo_loss) - float(loss_eval)) < 1e-5 * abs(float(loss_eval))
    # the 'TEDM' experiment-name defect (train_baseline.py:30 vs train.py:45): without replication the reference raises
    try:
        torch.nn.functional.binary_cross_entropy_with_logits(logits, y)
        raised = False
    except ValueError:
        raised = True
    assert raised
    # train mode: BatchNorm on the statistics of the (B*T) pixel batch, classifier gradients (train_datasetDM.py:46: only the head trains)
    torch.set_grad_enabled(True)
    model.train()
    model.classifier.zero_grad()
    logits_tr = model.classifier(feats)
    loss_tr = reduce(torch.nn.functional.binary_cross_entropy_with_logits(logits_tr, y_rep, reduction='none'), 'b c h w -> b c', 'mean').mean()
    loss_tr.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.classifier.named_parameters()}
    torch.set_grad_enabled(False)
    lt = O.tedm_classifier(pc, f_o, T, training=True)
    e = rel(lt, logits_tr)
    worst = max(worst, e)
    print(f'TEDM train-mode logits: oracle vs reference rel-L2 {e:.3e}; loss eval {float(loss_eval):.6f} train {float(loss_tr):.6f}; '
          f'dice {met["dice"].flatten().tolist()}')
    np.savez(GOLD / 'tedm_shared_d64_s64.npz', x0=x0.numpy(), noise=noise.numpy(), y=y.numpy(), steps=np.array(TEDM_STEPS),
             logits=logits.numpy(), prob=prob.numpy(), mask=mask.numpy(), loss_eval=np.float32(loss_eval),
             loss_per_t=loss_per_t.numpy(), **{k: v.numpy() for k, v in met.items()}, **{k: v.numpy() for k, v in met_t.items()},
             logits_train=logits_tr.numpy(), loss_train=np.float32(loss_tr),
             **{'bn:' + k: v for k, v in bn.items()},
             **{'grad:' + k: (v[:, ::8].contiguous().numpy() if k == '1.weight' else v.numpy()) for k, v in grads.items()},
             grad_norm_1w=np.float32(grads['1.weight'].norm()))

    # ---- 2. configs[2] at fixture size: 5 default timesteps, default width, 16x16 -----------------------------------
    steps5 = [50, 200, 400, 600, 800]                                     # config.py:78
    dm5 = RDM(cfg_ns(64, t_steps_to_save=steps5))
    dm5.diffusion_model.model.load_state_dict(w64)
    x0, _, noise = inputs(2, 16, 557)
    f_ref = dm5.extract_features(x0, noise=noise)
    assert tuple(f_ref.shape) == (2, 4800, 16, 16)
    e = rel(O.extract_features(w64, bufs, x0, steps5, noise), f_ref)
    worst = max(worst, e)
    print(f'd64 s16 extract_features T=5 {tuple(f_ref.shape)}: oracle vs reference rel-L2 {e:.3e}')
    np.savez(GOLD / 'extract5_d64_s16.npz', x0=x0.numpy(), noise=noise.numpy(), steps=np.array(steps5),
             feats_sub=f_ref[:, ::21].contiguous().numpy(), feats_mean=f_ref.mean(dim=(2, 3)).numpy())

    # ---- 3. reference-written checkpoint -----------------------------------------------------------------------------
    torch.set_grad_enabled(True)
    torch.manual_seed(5)
    cfg = Namespace(normalize=True, dim=8, dim_mults=(1, 2), channels=1, timesteps=1000, beta_schedule='cosine',
                    objective='pred_noise', p2_loss_weight_gamma=0., p2_loss_weight_k=1., img_size=16, lr=1e-4,
                    weight_decay=0.0, device='cpu', log_dir=Path('logs/tedm_fixture'), experiment='img_only', seed=5)
    dmr = RDiff(cfg)
    opt = torch.optim.Adam(dmr.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)
    xb, _, _ = inputs(4, 16, 31)
    for s in range(2):
        opt.zero_grad()
        torch.manual_seed(200 + s)
        loss = dmr.train_step(xb)
        loss.backward()
        opt.step()
    path = GOLD / 'reference_checkpoint_d8.pt'
    ref_save(dmr, opt, cfg, path, 2)                                      # trainers/train_base_diffusion.py:164-170
    torch.set_grad_enabled(False)
    dmr.eval()
    xq, tq, nq = inputs(2, 16, 32)
    x_t, _ = dmr.forward_diffusion_model(xq * 2 - 1, tq, nq)
    eps = dmr.model(x_t, tq)
    w8 = {k: v.detach() for k, v in dmr.model.state_dict().items()}
    e = rel(O.unet_forward(w8, x_t, tq, dim=8, dim_mults=(1, 2)), eps)
    worst = max(worst, e)
    st = opt.state_dict()['state']
    np.savez(GOLD / 'reference_checkpoint_d8_probe.npz', x_t=x_t.numpy(), t=tq.numpy(), eps=eps.numpy(),
             n_params=np.int64(sum(p.numel() for p in dmr.parameters())),
             exp_avg_sum=np.float64(sum(float(v['exp_avg'].double().sum()) for v in st.values())),
             param_sum=np.float64(sum(float(p.double().sum()) for p in dmr.parameters())))
    print(f'reference checkpoint: {path.stat().st_size} bytes, eps oracle vs reference rel-L2 {e:.3e}')

    print(f'worst oracle-vs-reference rel-L2: {worst:.3e}')
    if worst > 1e-5:
        raise SystemExit('oracle restatement deviates from the reference by more than 1e-5')


if __name__ == '__main__':
    main()
