"""CPU: TEDM host logic (tedm_amd/tedm.py) against the reference's recorded vectors (oracle/make_golden_tedm.py ->
tests/golden/tedm_shared_d64_s64.npz): the oracle's shared-weights classifier, mean-over-steps ensembling, replicated-target BCE,
Dice / precision / recall; the head-layout detection of DatasetDM (incl. the einops layer the reference swaps in); the
head training loop's cadence with a stub model; the reference-WRITTEN checkpoint."""
import json
from argparse import Namespace

import numpy as np
import pytest
import torch
from torch import nn

from oracle import unet_ref as O
from tedm_amd import tedm
from tedm_amd.weights import synth_state_dict


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


def head_weights(g):
    c = tedm.shared_weights_classifier(3)
    sd = c.state_dict()
    sd.update(synth_state_dict({k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point}, seed=6))
    for k in g.files:
        if k.startswith('bn:'):
            sd[k[3:]] = torch.from_numpy(g[k])
    return sd


def test_state_dict_keys_of_the_shared_head_are_the_references():
    keys = list(tedm.shared_weights_classifier(8).state_dict())
    assert keys == ['1.weight', '1.bias', '3.weight', '3.bias', '3.running_mean', '3.running_var', '3.num_batches_tracked',
                    '4.weight', '4.bias', '6.weight', '6.bias', '6.running_mean', '6.running_var', '6.num_batches_tracked',
                    '7.weight', '7.bias']
    c = tedm.shared_weights_classifier(8)
    assert c[1].weight.shape == (128, 960, 1, 1) and c[7].weight.shape == (1, 32, 1, 1)
    x = torch.arange(2 * 6 * 1 * 1, dtype=torch.float32).reshape(2, 6, 1, 1)
    from einops.layers.torch import Rearrange
    ref = Rearrange('b (step act) h w -> (b step) act h w', step=3)
    assert torch.equal(tedm.StepsToBatch(3)(x), ref(x))
    assert tedm.is_steps_to_batch(ref) and not tedm.is_steps_to_batch(Rearrange('b c h w -> b h w c'))


def test_oracle_tedm_head_vs_reference_golden(golden_dir):
    g = np.load(golden_dir / 'tedm_shared_d64_s64.npz')
    shapes = {k: tuple(v) for k, v in json.loads((golden_dir / 'unet_d64_state_dict_keys.json').read_text()).items()}
    w = synth_state_dict(shapes, 0)
    steps = [int(s) for s in g['steps']]
    T = len(steps)
    f = O.extract_features(w, O.diffusion_buffers(), torch.from_numpy(g['x0']), steps, torch.from_numpy(g['noise']))
    pc = {k: v for k, v in head_weights(g).items() if v.dtype.is_floating_point}
    logits = O.tedm_classifier(pc, f, T)
    assert rel(logits, g['logits']) < 1e-5
    prob, mask = O.tedm_ensemble(logits, T)
    assert rel(prob, g['prob']) < 1e-5 and (mask.numpy() != g['mask']).mean() < 1e-3
    y = torch.from_numpy(g['y'])
    loss, _ = O.bce_loss(logits, O.replicate_targets(y, T))
    assert abs(float(loss) - float(g['loss_eval'])) < 1e-5
    lt = O.tedm_classifier(pc, f, T, training=True)
    assert rel(lt, g['logits_train']) < 1e-5


def test_ensemble_loss_and_metrics_vs_reference_golden(golden_dir):
    g = np.load(golden_dir / 'tedm_shared_d64_s64.npz')
    T = len(g['steps'])
    logits, y = torch.from_numpy(g['logits']), torch.from_numpy(g['y'])
    prob, mask = tedm.ensemble(logits, T)
    assert torch.equal(mask, torch.from_numpy(g['mask'])) and rel(prob, g['prob']) < 1e-6
    y_rep = tedm.replicate_targets(y, T)
    assert y_rep.shape[0] == logits.shape[0] and torch.equal(y_rep[0], y_rep[T - 1]) and torch.equal(y_rep[T], y[1])
    loss, expanded = tedm.bce_loss(logits, y_rep)
    assert abs(float(loss) - float(g['loss_eval'])) < 1e-6
    per_t = expanded.reshape(-1, T, 1).mean(dim=(0, 2))
    assert np.allclose(per_t.numpy(), g['loss_per_t'], rtol=1e-5)
    for name, fn in (('dice', tedm.dice), ('precision', tedm.precision), ('recall', tedm.recall)):
        assert np.allclose(fn(mask, y).numpy(), g[name], rtol=1e-6, equal_nan=True), name
    pt = tedm.per_step_probabilities(logits, T)
    for i in range(T):
        assert np.allclose(tedm.dice(pt[i] > .5, y).numpy(), g[f'dice_t{i}'], rtol=1e-6, equal_nan=True)
    with pytest.raises(ValueError):                                    # the reference's 'TEDM' defect: un-replicated targets
        torch.nn.functional.binary_cross_entropy_with_logits(logits, y)
    empty = torch.zeros(1, 1, 4, 4, dtype=torch.bool)                  # 0/0 -> nan, dropped by nanmean (train_baseline.py:140)
    assert torch.isnan(tedm.dice(empty, empty)).all()


def test_head_layout_detection():
    from tedm_amd import DatasetDM
    base = dict(normalize=True, dim=32, saved_diffusion_model='/nonexistent', verbose=False, device='cpu', t_steps_to_save=[50, 150, 250])
    m = DatasetDM(Namespace(**base))
    assert m.feature_channels == 480 and m.head_layout() == ('concat', 0) and not tedm.has_shared_head(m)
    m = DatasetDM(Namespace(shared_weights_over_timesteps=True, **base))
    assert m.head_layout() == ('shared', 1) and tedm.has_shared_head(m) and m.classifier[1].in_channels == 480
    from einops.layers.torch import Rearrange
    m.classifier = nn.Sequential(Rearrange('b (step act) h w -> (b step) act h w', step=3), nn.Conv2d(480, 128, 1), nn.ReLU(),
                                 nn.BatchNorm2d(128), nn.Conv2d(128, 32, 1), nn.ReLU(), nn.BatchNorm2d(32), nn.Conv2d(32, 1, 1))
    assert m.head_layout() == ('shared', 1)                            # the swap trainers/train_datasetDM.py:33-42 performs
    m.classifier[0] = Rearrange('b (step act) h w -> (b step) act h w', step=2)
    assert m.head_layout() is None                                     # step count does not match model.steps: unfused path
    m.classifier = nn.Sequential(nn.Conv2d(480 * 3, 100, 1))
    assert m.head_layout() is None                                     # Cout % 16 != 0: no fused kernel


class StubDM(nn.Module):
    """stands in for DatasetDM on the CPU: per-pixel linear 'features' for T steps, then the shared head"""

    def __init__(self, T=3):
        super().__init__()
        self.steps = list(range(T))
        self.classifier = nn.Sequential(tedm.StepsToBatch(T), nn.Conv2d(2, 1, 1))

    def forward(self, x):
        f = torch.cat([torch.cat([x * (s + 1), 1 - x], 1) for s in self.steps], 1)
        return self.classifier(f)


def test_train_head_loop_replicates_targets_and_checkpoints(tmp_path):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(8, 1, 8, 8, generator=g)
    y = (x > 0.5).float()
    data = [(x[i:i + 4], y[i:i + 4]) for i in (0, 4)]
    cfg = Namespace(device='cpu', log_freq=5, val_freq=10, max_steps=60, max_val_steps=-1, debug=False, lr=5e-2,
                    log_dir=str(tmp_path / 'run'), early_stop=False, experiment='TEDM')
    m = StubDM()
    opt = torch.optim.Adam(m.classifier.parameters(), lr=cfg.lr)
    seen, logs = [], []

    class Logger:
        def log(self, d, step):
            logs.append((step, sorted(d)))
    first = tedm.validate_head(cfg, m, data)
    tedm.train_head(cfg, m, opt, data, data, Logger(), on_step=lambda s, l: seen.append(l))
    last = tedm.validate_head(cfg, m, data)
    assert len(seen) == 60 and seen[-1] < seen[0] and last['val/loss'] < first['val/loss'] and last['val/dice_ensemble'] > 0.9
    assert any(k == ['train_loss/step_0'] for _, k in logs)            # per-timestep loss logging (train_baseline.py:56-58,76-79)
    ck = torch.load(tmp_path / 'run' / 'best_model.pt', weights_only=False)
    assert sorted(ck) == ['config', 'model_state_dict', 'optimizer_state_dict', 'step'] and ck['config'].experiment == 'TEDM'
    m.eval()
    p, mask = tedm.predict(m, x[:2])
    assert p.shape == (2, 1, 8, 8) and mask.dtype == torch.bool and not m.training


def test_reference_written_checkpoint_loads(golden_dir):
    """a checkpoint WRITTEN BY THE REFERENCE (trainers/train_base_diffusion.py:164-170 `save`; Namespace + PosixPath
    inside, Adam state after two steps) loads through checkpoint.load_checkpoint (weights_only) and trainer.load into the
    build's modules, and the weights it carries reproduce the eps the reference computed from them (via the oracle)."""
    import pathlib
    from tedm_amd import trainer
    from tedm_amd.checkpoint import load_checkpoint
    path = golden_dir / 'reference_checkpoint_d8.pt'
    probe = np.load(golden_dir / 'reference_checkpoint_d8_probe.npz')
    ck = load_checkpoint(path)
    assert sorted(ck) == ['config', 'model_state_dict', 'optimizer_state_dict', 'step'] and ck['step'] == 2
    assert isinstance(ck['config'], Namespace) and isinstance(ck['config'].log_dir, pathlib.PurePath) and ck['config'].dim == 8
    dm, opt, step = trainer.load(Namespace(device='cpu', lr=1e-4), path)
    assert step == 2 and type(dm).__module__ == 'tedm_amd.diffusion'
    assert sum(p.numel() for p in dm.parameters()) == int(probe['n_params'])
    assert abs(sum(float(p.double().sum()) for p in dm.parameters()) - float(probe['param_sum'])) < 1e-9
    st = opt.state_dict()['state']
    assert len(st) == len(list(dm.parameters())) and all(int(v['step']) == 2 for v in st.values())
    assert abs(sum(float(v['exp_avg'].double().sum()) for v in st.values()) - float(probe['exp_avg_sum'])) < 1e-9
    w = {k: v.detach() for k, v in dm.model.state_dict().items()}
    eps = O.unet_forward(w, torch.from_numpy(probe['x_t']), torch.from_numpy(probe['t']), dim=8, dim_mults=(1, 2))
    assert rel(eps, probe['eps']) < 1e-6
    with pytest.raises(FileNotFoundError):                             # not reported as an unpickling problem
        load_checkpoint(golden_dir / 'no_such_file.pt')


def test_classifier_tail_kernel_dispatch_rules():
    """tedm_amd/head.py: the HIP tail kernels take over exactly for the reference's classifier tail (models/datasetDM_model.py:59-64)
    -- folded single kernel in eval mode, six-pass kernels in training mode -- and leave every other tail to the module's own layers."""
    from torch import nn
    from tedm_amd.head import eval_tail_supported, train_tail_supported

    def tail(c1=128, mid=32, momentum=0.1, affine=True, track=True, bias=True):
        return [nn.ReLU(), nn.BatchNorm2d(c1, momentum=momentum, affine=affine, track_running_stats=track), nn.Conv2d(c1, mid, 1, bias=bias),
                nn.ReLU(), nn.BatchNorm2d(mid, momentum=momentum, affine=affine, track_running_stats=track), nn.Conv2d(mid, 1, 1, bias=bias)]

    t = tail()
    assert train_tail_supported(t, 128) and not eval_tail_supported(t, 128)              # fresh modules are in training mode
    for m in t:
        m.eval()
    assert eval_tail_supported(t, 128) and not train_tail_supported(t, 128)
    assert train_tail_supported(tail(256), 256)
    assert not train_tail_supported(tail(64), 64)                     # width the gradient kernel is not built for -> torch layers
    assert not train_tail_supported(tail(momentum=None), 128)         # cumulative-average BatchNorm: torch layers
    assert not train_tail_supported(tail(affine=False), 128)
    assert not train_tail_supported(tail(track=False), 128)
    assert not train_tail_supported(tail(bias=False), 128)
    assert not train_tail_supported(tail(mid=16), 128)
    assert not train_tail_supported(tail()[:-1] + [nn.Conv2d(32, 2, 1)], 128)
    assert not train_tail_supported(tail() + [nn.Sigmoid()], 128)
    assert not train_tail_supported(tail(), 256)                      # first-layer width does not match the BatchNorm


def test_reference_written_checkpoint_d32_loads(golden_dir):
    """The reference-written checkpoint the HIP kernels accept (oracle/make_golden_ckpt.py: dim = 32, dim_mults = (1, 2)) on the
    CPU: loader, module tree (every key, no missing / unexpected), and the oracle reproducing the eps the reference computed."""
    from tedm_amd import trainer
    probe = np.load(golden_dir / 'reference_checkpoint_d32_probe.npz')
    dm, opt, step = trainer.load(Namespace(device='cpu', lr=1e-4), golden_dir / 'reference_checkpoint_d32.pt')
    assert step == 2 and sum(p.numel() for p in dm.parameters()) == int(probe['n_params'])
    assert abs(sum(float(p.double().sum()) for p in dm.parameters()) - float(probe['param_sum'])) < 1e-9
    w = {k: v.detach() for k, v in dm.model.state_dict().items()}
    eps = O.unet_forward(w, torch.from_numpy(probe['x_t']), torch.from_numpy(probe['t']), dim=32, dim_mults=(1, 2))
    assert rel(eps, probe['eps']) < 1e-6
