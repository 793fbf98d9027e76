"""GPU (MI355X): TEDM proper (BASELINE configs[4]) -- the SHARED-WEIGHTS pixel classifier over timesteps with its
first layer fused into the decoder taps, against the reference's recorded vectors (tests/golden/tedm_shared_d64_s64.npz from
oracle/make_golden_tedm.py: eval logits, mean-over-steps mask, replicated-target BCE, train-mode logits and classifier
gradients), the unfused formulation, and a JSRT-shaped 128x128 / 8-timestep end-to-end run (held-out Dice)."""
from argparse import Namespace

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope='module')
def dev(built_lib):
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


def make(dev, golden_dir, prec, via_swap=False):
    from tedm_amd import DatasetDM, load_synth_weights, tedm
    from tedm_amd.weights import synth_state_dict
    g = np.load(golden_dir / 'tedm_shared_d64_s64.npz')
    steps = [int(s) for s in g['steps']]
    args = Namespace(normalize=True, dim=64, saved_diffusion_model='/nonexistent', verbose=False, device='cuda',
                     t_steps_to_save=steps, shared_weights_over_timesteps=not via_swap, precision=prec, train_precision=prec)
    dm = DatasetDM(args)
    if via_swap:                                            # what trainers/train_datasetDM.py:30-42 / app.py:66-75 do to a built model
        from einops.layers.torch import Rearrange
        dm.classifier = nn.Sequential(Rearrange('b (step act) h w -> (b step) act h w', step=len(dm.steps)), nn.Conv2d(960, 128, 1),
                                      nn.ReLU(), nn.BatchNorm2d(128), nn.Conv2d(128, 32, 1), nn.ReLU(), nn.BatchNorm2d(32),
                                      nn.Conv2d(32, 1, 1))
    assert dm.head_layout() == ('shared', 1) and tedm.has_shared_head(dm)
    load_synth_weights(dm.diffusion_model.model, seed=0)
    sd = dm.classifier.state_dict()
    sd.update(synth_state_dict({k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point}, seed=6))
    for k in g.files:
        if k.startswith('bn:'):
            sd[k[3:]] = torch.from_numpy(g[k])
    dm.classifier.load_state_dict(sd)
    return dm.to(dev), g


@pytest.mark.parametrize('prec,via_swap', [('bf16x3', False), ('f16f4', False), ('bf16x3', True)])
def test_shared_head_eval_vs_reference_golden(dev, golden_dir, prec, via_swap):
    from tedm_amd import tedm
    dm, g = make(dev, golden_dir, prec, via_swap)
    dm.eval()
    T = len(dm.steps)
    x0, noise, y = (torch.from_numpy(g[k]).to(dev) for k in ('x0', 'noise', 'y'))
    with torch.no_grad():
        fused = dm(x0, noise=noise)                          # projections + ONE tail kernel (head.fused_head_eval)
        dm.fused_tail = False
        first_only = dm(x0, noise=noise)                     # fused first layer, torch tail (what training runs)
        dm.fused_tail = True
        dm.fused_head = False
        plain = dm(x0, noise=noise)
        dm.fused_head = True
    assert tuple(fused.shape) == tuple(g['logits'].shape) == (2 * T, 1, 64, 64)
    assert rel(fused, first_only) < 1e-5                     # folded BatchNorms / fp32 VALU tail vs torch's layers
    tol = 2e-3 if prec == 'bf16x3' else 2e-2               # the random head amplifies the tap error (2e-5 / 3e-4) ~50x
    e_ref, e_plain = rel(fused, g['logits']), rel(fused, plain)
    prob, mask = tedm.ensemble(fused, T)
    flips = float((mask.cpu().numpy() != g['mask']).mean())
    loss, _ = tedm.bce_loss(fused, tedm.replicate_targets(y, T))
    d = tedm.dice(mask, y).cpu().numpy()
    print(f'TEDM shared head {prec}{" (einops swap)" if via_swap else ""}: logits vs reference {e_ref:.2e}, vs unfused {e_plain:.2e}, '
          f'mask flips {flips:.2e}, loss {float(loss):.6f} (ref {float(g["loss_eval"]):.6f}), dice {d.ravel()} (ref {g["dice"].ravel()})')
    assert e_ref < tol and e_plain < tol
    assert rel(prob, g['prob']) < tol and flips < (2e-3 if prec == 'bf16x3' else 2e-2)
    assert abs(float(loss) - float(g['loss_eval'])) < tol * float(g['loss_eval'])
    assert np.allclose(d, g['dice'], atol=5e-3 if prec == 'bf16x3' else 3e-2)
    p2, m2 = tedm.predict(dm, x0, noise=noise)              # app.py:62-79 on a batch
    assert torch.equal(m2, mask)
    one = tedm.predict(dm, x0[:1], noise=noise[:1])[1]      # and on one image: sigmoid(TEDM(img)).mean(0) > .5
    assert float((one != mask[:1]).float().mean()) < 1e-3


def test_shared_head_train_mode_logits_and_gradients_vs_reference(dev, golden_dir):
    """BatchNorm on the statistics of the (B*T) pixel batch, replicated targets, BCE (trainers/train_baseline.py:30,44-45),
    gradients of every classifier parameter against the reference's autograd; the frozen backbone gets none."""
    from tedm_amd import tedm
    dm, g = make(dev, golden_dir, 'bf16x3')
    dm.train()
    dm.diffusion_model.eval()
    T = len(dm.steps)
    x0, noise, y = (torch.from_numpy(g[k]).to(dev) for k in ('x0', 'noise', 'y'))
    res = {}
    for fused in (True, False):
        dm.fused_head = fused
        dm.classifier.zero_grad()
        for m in dm.classifier:                              # same starting statistics for both passes
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.from_numpy(g[f'bn:{3 if m.num_features == 128 else 6}.running_mean']))
                m.running_var.copy_(torch.from_numpy(g[f'bn:{3 if m.num_features == 128 else 6}.running_var']))
        logits = dm(x0, noise=noise)
        loss, _ = tedm.bce_loss(logits, tedm.replicate_targets(y, T))
        loss.backward()
        res[fused] = (logits.detach(), float(loss), {k: p.grad.clone() for k, p in dm.classifier.named_parameters()})
    logits, loss, grads = res[True]
    print(f'train-mode logits vs reference {rel(logits, g["logits_train"]):.2e}, loss {loss:.6f} (ref {float(g["loss_train"]):.6f})')
    assert rel(logits, g['logits_train']) < 2e-3 and abs(loss - float(g['loss_train'])) < 2e-3 * float(g['loss_train'])
    assert rel(res[False][0], logits) < 1e-3
    worst = 0.0
    for k, gr in grads.items():
        ref = g['grad:' + k]
        mine = gr[:, ::8] if k == '1.weight' else gr
        e, e2 = rel(mine, ref), rel(gr, res[False][2][k])
        worst = max(worst, e, e2)
        assert e < 1e-2 and e2 < 1e-2, (k, e, e2)
    assert abs(float(grads['1.weight'].norm()) - float(g['grad_norm_1w'])) < 1e-2 * float(g['grad_norm_1w'])
    print(f'classifier gradients: worst rel-L2 vs reference / unfused autograd {worst:.2e}')
    assert all(p.grad is None for p in dm.diffusion_model.parameters())


def test_shared_head_chunked_forward_equals_single(dev, golden_dir):
    """`max_images_per_forward` cuts the (b step) batch along b: the taps and the fused 960 -> 128 projection are the same bit for
    bit (no cross-sample op in the UNet; the tail of the head is torch's own conv / BatchNorm kernels, held to 1e-6)."""
    from tedm_amd.head import fused_first_layer
    dm, g = make(dev, golden_dir, 'bf16x3')
    dm.eval()
    x0, noise = torch.from_numpy(g['x0']).to(dev), torch.from_numpy(g['noise']).to(dev)
    with torch.no_grad():
        a = dm(x0, noise=noise)
        ta = [t.clone() for t in dm._taps_steps_in_batch(x0, noise)]
        ha = fused_first_layer(dm.classifier[1], [ta], 64, 0)
        dm.max_images_per_forward = 3                        # one image (T = 3 forwards) per native call
        b = dm(x0, noise=noise)
        tb = dm._taps_steps_in_batch(x0, noise)
        hb = fused_first_layer(dm.classifier[1], [tb], 64, 0)
    assert all(torch.equal(p, q) for p, q in zip(ta, tb)) and torch.equal(ha, hb)
    assert rel(b, a) < 1e-6


def test_tedm_end_to_end_jsrt_shaped_128(dev):
    """BASELINE configs[4] at its shape: frozen diffusion UNet, 128x128 single-channel images + binary lung-like masks, the 8 TEDM
    timesteps (train.py:47), shared-weights classifier trained with Adam/BCE on replicated targets through tedm.train_head;
    held-out Dice of the mean-over-steps mask > 0.8 (examples/tedm_synthetic.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'examples'))
    import tedm_synthetic
    hist = tedm_synthetic.run(steps=60, size=128, n_train=24, n_val=8, batch=8, lr=1e-3, verbose=False)
    first, last = hist[0], hist[-1]
    print(f"TEDM 128x128, 8 timesteps: val loss {first['val/loss']:.3f} -> {last['val/loss']:.3f}, held-out ensemble Dice "
          f"{first['val/dice_ensemble']:.3f} -> {last['val/dice_ensemble']:.3f}, per-step Dice {last['val/dice']:.3f}")
    assert last['val/loss'] < first['val/loss'] and last['val/dice_ensemble'] > 0.8
