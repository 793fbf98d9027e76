"""CPU: the C-ABI library loads and exports every symbol the headers declare; the facade keeps the
reference's state-dict contract; host-side plan / workspace logic; no compute is launched."""
import ctypes as C
import json
import re
from argparse import Namespace

import pytest
import torch

from oracle import unet_ref as O


def header_functions(root):
    names = []
    for h in sorted((root / 'include').glob('*.h')):
        text = re.sub(r'/\*.*?\*/', '', h.read_text(), flags=re.S)
        names += re.findall(r'\b(tedm_[a-z0-9_]+)\s*\(', text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(built_lib):
    from tests.conftest import ROOT
    lib = C.CDLL(str(built_lib))
    declared = header_functions(ROOT)
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/ but not exported'
    from tedm_amd import _lib
    assert set(_lib.EXPORTED_SYMBOLS) == set(declared)


def test_library_is_built_from_the_sources_next_to_it(built_lib, monkeypatch):
    """The binary reports the digest of the sources it was compiled from; the loader compares it with the tree and refuses a
    stale library instead of running old kernels under new host code (the .so is git-ignored and travels to the GPU box as built)."""
    from tedm_amd import _lib, build
    assert build.library_digest(built_lib) == build.source_digest()
    assert not build.needs_build()
    monkeypatch.setattr(build, 'source_digest', lambda: 'f' * 64)            # "someone edited a kernel and did not rebuild"
    assert build.needs_build()
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(ImportError, match='built from other sources'):
        _lib.lib()
    monkeypatch.setenv('TEDM_ALLOW_STALE_LIB', '1')
    assert _lib.lib() is not None


def test_state_dict_contract(golden_dir):
    from tedm_amd import Unet
    want = json.loads((golden_dir / 'unet_d64_state_dict_keys.json').read_text())
    sd = Unet(64, dim_mults=(1, 2, 4, 8), channels=1).state_dict()
    assert list(sd.keys()) == list(want.keys())
    assert all(list(sd[k].shape) == want[k] for k in want)
    assert sum(v.numel() for v in sd.values()) == 36245377


def test_native_plan_matches_facade(built_lib):
    from tedm_amd import Unet, _lib
    L = _lib.lib()
    for dim in (64, 32):
        m = Unet(dim)
        cfg = _lib.UnetCfg()
        cfg.dim, cfg.n_mults, cfg.channels, cfg.groups, cfg.heads, cfg.dim_head = dim, 4, 1, 8, 4, 32
        for i, v in enumerate((1, 2, 4, 8)):
            cfg.dim_mults[i] = v
        h = C.c_void_p()
        _lib.check(L.tedm_unet_create(C.byref(cfg), C.byref(h)))
        n = L.tedm_unet_num_params(h)
        native = {L.tedm_unet_param_key(h, i).decode(): L.tedm_unet_param_numel(h, i) for i in range(n)}
        facade = {k: v.numel() for k, v in m._native_tensors().items()}
        assert native == facade
        assert L.tedm_unet_workspace_bytes(h, 32, 256, 256, 0) > (1 << 30)
        assert L.tedm_unet_workspace_bytes(h, 2, 60, 64, 0) == 0          # H not divisible by 8
        assert L.tedm_unet_workspace_bytes(h, 2, 64, 64, _lib.FWD_TAPS_ONLY) <= L.tedm_unet_workspace_bytes(h, 2, 64, 64, 0)
        L.tedm_unet_destroy(h)


def test_bad_config_is_rejected(built_lib):
    from tedm_amd import _lib
    L = _lib.lib()
    cfg = _lib.UnetCfg()
    cfg.dim, cfg.n_mults, cfg.channels, cfg.groups, cfg.heads, cfg.dim_head = 24, 4, 1, 8, 4, 32
    h = C.c_void_p()
    assert L.tedm_unet_create(C.byref(cfg), C.byref(h)) != 0
    assert b'multiple of 32' in L.tedm_last_error()


def test_no_cpu_fallback():
    from tedm_amd import DiffusionModel, Unet
    m = Unet(32)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        with torch.no_grad():
            m(torch.zeros(1, 1, 16, 16), torch.zeros(1))
    with pytest.raises(RuntimeError, match='no CPU fallback'):        # stand-alone block call: native too
        m.downs[0][0](torch.zeros(1, 32, 16, 16))
    with pytest.raises(NotImplementedError):                          # inner containers stay containers
        m.downs[0][0].block1(torch.zeros(1, 32, 16, 16))
    import copy
    assert len(copy.deepcopy(m).state_dict()) == 276                   # the owner back-reference survives deepcopy (EMA copies)
    cfg = Namespace(normalize=True, dim=32)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        DiffusionModel(cfg).forward_diffusion_model(torch.zeros(1, 1, 16, 16), torch.zeros(1, dtype=torch.long))


def test_diffusion_buffers_match_oracle():
    from tedm_amd import DiffusionModel
    for sched in ('cosine', 'linear'):
        dm = DiffusionModel(Namespace(normalize=True, dim=32, beta_schedule=sched, p2_loss_weight_gamma=1.0))
        bufs = O.diffusion_buffers(1000, sched, p2_gamma=1.0)
        for k, v in bufs.items():
            assert torch.equal(getattr(dm, k), v), k
    with pytest.raises(ValueError):
        DiffusionModel(Namespace(normalize=True, dim=32, beta_schedule='nope'))


def test_synth_weights_deterministic():
    from tedm_amd import Unet, load_synth_weights, synth_state_dict
    a, b = Unet(32), Unet(32)
    load_synth_weights(a, seed=3)
    load_synth_weights(b, seed=3)
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    s = synth_state_dict({k: tuple(v.shape) for k, v in a.state_dict().items()}, 3)
    assert all(torch.equal(s[k], a.state_dict()[k]) for k in s)
    load_synth_weights(b, seed=4)
    assert not torch.equal(a.init_conv.weight, b.init_conv.weight)


def test_checkpoint_roundtrip(tmp_path):
    """Reference checkpoint format (trainers/train_CXR14.py:96-102) loads into the facade."""
    from tedm_amd import DiffusionModel, load_synth_weights
    cfg = Namespace(normalize=True, dim=32, dim_mults=(1, 2, 4, 8), channels=1)
    dm = DiffusionModel(cfg)
    load_synth_weights(dm.model, seed=1)
    path = tmp_path / 'best_model.pt'
    torch.save({'model_state_dict': dm.state_dict(), 'optimizer_state_dict': {}, 'config': cfg, 'step': 5}, path)
    ck = torch.load(path, weights_only=False)
    dm2 = DiffusionModel(ck['config'])
    dm2.load_state_dict(ck['model_state_dict'])
    assert all(torch.equal(a, b) for a, b in zip(dm.state_dict().values(), dm2.state_dict().values()))


def test_workspace_sizing_is_host_only_and_monotonic(built_lib):
    """tedm_unet_workspace_bytes / tedm_unet_train_workspace_bytes are dry runs of the launch sequence (no device)."""
    from tedm_amd import _lib
    L = _lib.lib()
    cfg = _lib.UnetCfg()
    cfg.dim, cfg.n_mults, cfg.channels, cfg.groups, cfg.heads, cfg.dim_head = 64, 4, 1, 8, 4, 32
    for i, v in enumerate((1, 2, 4, 8)):
        cfg.dim_mults[i] = v
    h = C.c_void_p()
    _lib.check(L.tedm_unet_create(C.byref(cfg), C.byref(h)))
    f1, f2 = L.tedm_unet_workspace_bytes(h, 2, 64, 64, 0), L.tedm_unet_workspace_bytes(h, 4, 64, 64, 0)
    t1, t2 = L.tedm_unet_train_workspace_bytes(h, 2, 64, 64), L.tedm_unet_train_workspace_bytes(h, 4, 64, 64)
    assert 0 < f1 < f2 and 0 < t1 < t2 and t1 > f1              # the tape keeps what the inference arena recycles
    assert L.tedm_unet_workspace_bytes(h, 2, 60, 64, 0) == 0    # H must be a multiple of 8
    assert L.tedm_unet_train_workspace_bytes(h, 0, 64, 64) == 0
    # training entry points refuse to run before parameters / gradients are bound and packed
    assert L.tedm_unet_forward_train(h, C.c_void_p(8), None, 2, 64, 64, C.c_void_p(8), C.c_void_p(8), 1 << 20, None, 0) != 0
    assert b'pack' in L.tedm_last_error()
    assert L.tedm_unet_backward(h, C.c_void_p(8), C.c_void_p(8), 1 << 20, None) != 0
    assert b'tape' in L.tedm_last_error()
    L.tedm_unet_destroy(h)


def test_learned_sinusoidal_cond_contract(golden_dir, built_lib):
    """`Unet(learned_sinusoidal_cond=True)` (models/unet_model.py:96-114,278-285): the state-dict keys of the reference (the learned
    frequency vector `time_mlp.0.weights`, a 17-input `time_mlp.1`), the native plan asking for exactly these tensors, and the
    oracle reproducing the eps the reference computed (fixture: oracle/make_golden_ckpt.py --learned)."""
    import numpy as np
    import torch
    from oracle import unet_ref as O
    from tedm_amd import Unet, load_synth_weights
    g = np.load(golden_dir / 'learned_sinusoid_d32_s32.npz')
    m = Unet(32, dim_mults=(1, 2), channels=1, learned_sinusoidal_cond=True, learned_sinusoidal_dim=16)
    assert sorted(m.state_dict()) == [str(k) for k in g['keys']]
    assert m.time_mlp[1].in_features == int(g['fourier_in']) == 17 and tuple(m.time_mlp[0].weights.shape) == (8,)
    load_synth_weights(m, seed=3)
    w = {k: v.detach() for k, v in m.state_dict().items()}
    eps = O.unet_forward(w, torch.from_numpy(g['x']), torch.from_numpy(g['t']), dim=32, dim_mults=(1, 2))
    assert float((eps - torch.from_numpy(g['eps'])).norm() / torch.from_numpy(g['eps']).norm()) < 1e-6
