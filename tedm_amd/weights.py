"""Deterministic synthetic weights.

No trained TEDM checkpoint ships with the reference (the default
`logs/.../best_model.pt`, config.py:77, is absent), so parity and benchmark runs use a
build-owned generator: numpy PCG64 streams keyed by (seed, crc32(state-dict key)), fan-in
scaled.  The same call regenerates bit-identical weights in this container (where the
goldens are made against the imported reference) and on the GPU box.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Tuple

import numpy as np
import torch


def _gen(key: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'g':                                   # channel-LayerNorm gain (1,C,1,1)
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if '.norm.' in key and leaf == 'weight':          # GroupNorm gamma
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == 'bias':                                # GroupNorm beta, conv / linear bias
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == 'weight':                              # conv (O,I,kh,kw) / linear (O,I)
        fan_in = int(np.prod(shape[1:]))
        return (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    if leaf in ('running_mean', 'num_batches_tracked'):
        return np.zeros(shape, dtype=np.float32 if leaf == 'running_mean' else np.int64)
    if leaf == 'running_var':
        return np.ones(shape, dtype=np.float32)
    return (0.1 * rng.standard_normal(shape)).astype(np.float32)


def synth_state_dict(shapes: Mapping[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """shapes: state-dict key -> shape (e.g. {k: tuple(v.shape) for k, v in m.state_dict().items()})."""
    return {k: torch.from_numpy(_gen(k, tuple(s), seed)) for k, s in shapes.items()}


def load_synth_weights(module: torch.nn.Module, seed: int = 0, prefix_filter: str = '') -> None:
    """Overwrite every floating-point tensor of module.state_dict() whose key starts with
    `prefix_filter` with the deterministic stream for its key (key taken relative to the module)."""
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()
              if k.startswith(prefix_filter) and v.dtype.is_floating_point}
    new = synth_state_dict(shapes, seed)
    with torch.no_grad():
        for k, v in new.items():
            sd[k].copy_(v.to(sd[k].device))
