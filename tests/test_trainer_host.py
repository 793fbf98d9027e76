"""CPU: host logic of the training-step caller (tedm_amd/trainer.py; reference trainers/train_CXR14.py:16-114) --
loop order, logging/validation cadence, best-validation checkpoint in the reference's dict format, resume.
The HIP compute path is not involved: a stub module with `train_step` stands in for the diffusion model."""
from argparse import Namespace

import pytest
import torch
from torch import nn


class Stub(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        self.config = config
        self.w = nn.Parameter(torch.tensor([2.0, -1.0]))
        self.register_buffer('betas', torch.linspace(0, 1, 5))

    def train_step(self, x, cond=None, t=None):
        return ((x * self.w).sum(dim=1) - 1.0).abs().mean()


def cfg(tmp_path, **kw):
    base = dict(device='cpu', log_freq=2, val_freq=3, max_steps=7, max_val_steps=2, debug=False, lr=1e-2,
                log_dir=str(tmp_path / 'run'), resume_path=None)
    base.update(kw)
    return Namespace(**base)


def loaders():
    g = torch.Generator().manual_seed(0)
    train = [torch.rand(4, 2, generator=g) for _ in range(3)]                 # CXR14 style: x only
    val = [(torch.rand(4, 2, generator=g), torch.zeros(4)) for _ in range(4)]  # JSRT style: (x, y)
    return train, val


def test_loop_cadence_checkpoint_and_resume(tmp_path):
    from tedm_amd import trainer
    c = cfg(tmp_path)
    m = Stub(c)
    opt = torch.optim.Adam(m.parameters(), lr=c.lr)
    tr, va = loaders()
    logs, steps = [], []

    class Logger:
        def log(self, d, step):
            logs.append((step, sorted(d)))
    out = trainer.train(c, m, opt, tr, va, Logger(), step=0, on_step=lambda s, l: steps.append((s, l)))
    assert out is m
    assert [s for s, _ in steps] == list(range(1, 8))                          # epochs wrap around the 3-batch loader
    assert [s for s, k in logs if k == ['train/loss']] == [2, 4, 6]
    assert [s for s, k in logs if k == ['val/loss']] == [3, 6]
    ck = torch.load(tmp_path / 'run' / 'best_model.pt', weights_only=False)
    assert sorted(ck) == ['config', 'model_state_dict', 'optimizer_state_dict', 'step']        # train_CXR14.py:96-102
    assert ck['step'] in (3, 6) and isinstance(ck['config'], Namespace)
    assert sorted(ck['model_state_dict']) == ['betas', 'w']
    m2, opt2, step2 = trainer.load(cfg(tmp_path, lr=5e-3), tmp_path / 'run' / 'best_model.pt', model_cls=Stub)
    assert step2 == ck['step'] and torch.equal(m2.w.detach(), ck['model_state_dict']['w'])
    assert opt2.state_dict()['state'][0]['step'] == ck['optimizer_state_dict']['state'][0]['step']
    # resumed training continues the step counter and stops at max_steps
    more = []
    trainer.train(cfg(tmp_path, max_steps=step2 + 2), m2, opt2, tr, None, None, step=step2, on_step=lambda s, l: more.append(s))
    assert more == [step2 + 1, step2 + 2]


def test_reference_format_checkpoint_roundtrip_of_the_diffusion_module(tmp_path):
    """a checkpoint in the reference's format (Namespace pickled inside) loads into the build's
    module; `model.` prefix, 276 UNet keys + 9 schedule buffers; byte-equal round trip."""
    from tedm_amd import DiffusionModel, trainer
    from tedm_amd.weights import load_synth_weights
    c = Namespace(normalize=True, dim=32, dim_mults=(1, 2, 4, 8), channels=1, timesteps=1000, beta_schedule='cosine',
                  objective='pred_noise', device='cpu', lr=1e-4)
    dm = DiffusionModel(c)
    load_synth_weights(dm.model, seed=2)
    opt = torch.optim.Adam(dm.parameters(), lr=c.lr)
    path = tmp_path / 'best_model.pt'
    trainer.save(dm, opt, c, path, 41)
    dm2, opt2, step = trainer.load(c, path)
    assert step == 41
    sd, sd2 = dm.state_dict(), dm2.state_dict()
    assert list(sd) == list(sd2) and len([k for k in sd if k.startswith('model.')]) == 276
    assert len([k for k in sd if not k.startswith('model.')]) == 9
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)


def test_checkpoint_loader_refuses_arbitrary_pickles(tmp_path):
    """Reference checkpoints hold an argparse Namespace (+ PosixPath): loaded with weights_only=True and those two types
    allow-listed; anything that needs full unpickling is refused unless the caller opts in."""
    import pathlib
    from tedm_amd.checkpoint import load_checkpoint
    good = tmp_path / 'good.pt'
    torch.save({'model_state_dict': {'w': torch.ones(3)}, 'config': Namespace(dim=64, log_dir=pathlib.Path('/tmp/x')), 'step': 7}, good)
    ck = load_checkpoint(good)
    assert ck['step'] == 7 and ck['config'].dim == 64 and torch.equal(ck['model_state_dict']['w'], torch.ones(3))

    class Evil:
        def __reduce__(self):
            return (print, ('arbitrary code ran',))
    bad = tmp_path / 'bad.pt'
    torch.save({'config': Evil()}, bad)
    with pytest.raises(RuntimeError, match='weights_only'):
        load_checkpoint(bad)
    assert load_checkpoint(bad, allow_unsafe=True) is not None     # explicit opt-in only
