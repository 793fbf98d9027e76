"""Loading reference-format checkpoints ({'model_state_dict', 'optimizer_state_dict', 'config': Namespace, 'step'},
trainers/train_CXR14.py:96-102) without unpickling arbitrary objects."""
from __future__ import annotations

import argparse
import os
import pathlib
import pickle
import warnings

import torch

_SAFE = [argparse.Namespace, pathlib.PosixPath, pathlib.PurePosixPath, pathlib.Path]


def load_checkpoint(path, map_location=None, allow_unsafe: bool = False):
    """torch.load(weights_only=True) with the two non-tensor types a reference checkpoint holds (the argparse Namespace under
    'config' and its PosixPath members) allow-listed.  A checkpoint that needs anything else is refused unless the caller
    opts in with `allow_unsafe=True` (the explicit argument; TEDM_UNSAFE_LOAD=1 is honoured for scripts that cannot pass it):
    unpickling can execute code.  Only the unpickler's refusal triggers that route -- a missing file, a corrupt archive or
    a bad map_location propagate as what they are."""
    try:
        with torch.serialization.safe_globals(_SAFE):
            return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:
        if allow_unsafe or os.environ.get('TEDM_UNSAFE_LOAD') == '1':
            warnings.warn(f'{path}: needs full unpickling ({str(e).splitlines()[0]}); loading with weights_only=False because '
                          f'{"allow_unsafe=True" if allow_unsafe else "TEDM_UNSAFE_LOAD=1"} -- this can execute code from the file',
                          RuntimeWarning, stacklevel=2)
            return torch.load(path, map_location=map_location, weights_only=False)
        raise RuntimeError(f'{path}: not loadable with weights_only=True (+ argparse.Namespace / pathlib paths): {e}\n'
                           'If you trust the file, pass allow_unsafe=True.') from e
