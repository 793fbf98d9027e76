"""GPU (MI355X): the pixel-classifier head with its first layer fused into the taps (tedm_amd/head.py)
against (i) the reference's recorded logits and (ii) the unfused path `classifier(extract_features(x))`, forward and
gradients (TEDM trains only the head: trainers/train_datasetDM.py:46)."""
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope='module')
def dev(built_lib):
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


def make(dev, golden_dir):
    from tedm_amd import DatasetDM, load_synth_weights
    g = np.load(golden_dir / 'classifier_d64_s16.npz')
    args = Namespace(normalize=True, dim=64, saved_diffusion_model='/nonexistent', verbose=False, device='cuda',
                     t_steps_to_save=[int(s) for s in g['steps']])
    dm = DatasetDM(args)
    load_synth_weights(dm.diffusion_model.model, seed=0)
    load_synth_weights(dm.classifier, seed=5)
    return dm.to(dev), g


def test_fused_head_logits_vs_reference_and_unfused(dev, golden_dir):
    dm, g = make(dev, golden_dir)
    dm.eval()
    x0, noise = torch.from_numpy(g['x0']).to(dev), torch.from_numpy(g['noise']).to(dev)
    with torch.no_grad():
        fused = dm(x0, noise=noise)                          # eval + no_grad: projections + the one-kernel tail
        dm.fused_tail = False
        first_only = dm(x0, noise=noise)
        dm.fused_head = False
        plain = dm(x0, noise=noise)
    assert tuple(fused.shape) == tuple(g['logits'].shape)
    assert rel(fused, first_only) < 1e-5
    assert rel(fused, plain) < 1e-4
    assert rel(fused, g['logits']) < 5e-3          # same bound as the unfused path (the head amplifies the 2e-5 tap error)


def test_fused_head_gradients_match_unfused_autograd(dev, golden_dir):
    dm, g = make(dev, golden_dir)
    dm.train()                                       # BatchNorm on batch statistics, as in train_datasetDM
    dm.diffusion_model.eval()
    x0, noise = torch.from_numpy(g['x0']).to(dev), torch.from_numpy(g['noise']).to(dev)
    target = (torch.rand(g['logits'].shape, generator=torch.Generator().manual_seed(3)) > 0.5).float().to(dev)
    grads = {}
    for fused in (True, False):
        dm.fused_head = fused
        dm.classifier.zero_grad()
        logits = dm(x0, noise=noise)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, target)    # train_baseline.py:44
        loss.backward()
        grads[fused] = {k: p.grad.clone() for k, p in dm.classifier.named_parameters()}
        grads[fused]['loss'] = loss.detach()
    assert abs(float(grads[True]['loss']) - float(grads[False]['loss'])) < 1e-4 * abs(float(grads[False]['loss']))
    for k in grads[False]:
        if k != 'loss':
            assert rel(grads[True][k], grads[False][k]) < 2e-3, (k, rel(grads[True][k], grads[False][k]))
    assert all(p.grad is None for p in dm.diffusion_model.parameters())           # the diffusion model stays frozen
