"""Loading reference-format checkpoints ({'model_state_dict', 'optimizer_state_dict', 'config': Namespace, 'step'},
trainers/train_CXR14.py:96-102) without unpickling arbitrary objects."""
from __future__ import annotations

import argparse
import os
import pathlib

import torch

_SAFE = [argparse.Namespace, pathlib.PosixPath, pathlib.PurePosixPath, pathlib.Path]


def load_checkpoint(path, map_location=None, allow_unsafe: bool = False):
    """torch.load(weights_only=True) with the two non-tensor types a reference checkpoint holds (the argparse Namespace under
    'config' and its PosixPath members) allow-listed.  A checkpoint that needs anything else is refused unless the caller
    opts in (`allow_unsafe=True` or TEDM_UNSAFE_LOAD=1): unpickling can execute code."""
    try:
        with torch.serialization.safe_globals(_SAFE):
            return torch.load(path, map_location=map_location, weights_only=True)
    except Exception as e:                                   # pickle.UnpicklingError and friends
        if allow_unsafe or os.environ.get('TEDM_UNSAFE_LOAD') == '1':
            return torch.load(path, map_location=map_location, weights_only=False)
        raise RuntimeError(f'{path}: not loadable with weights_only=True (+ argparse.Namespace / pathlib paths): {e}\n'
                           'If you trust the file, pass allow_unsafe=True or set TEDM_UNSAFE_LOAD=1.') from e
