"""GPU: BASELINE config 5 in miniature (examples/tedm_synthetic.py): frozen diffusion UNet + pixel-classifier head trained
with BCE on synthetic lung-like masks; the held-out Dice must go up (the reference's metric, trainers/train_baseline.py:146-161)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'examples'))


def test_tedm_head_learns_synthetic_masks(built_lib):
    import tedm_synthetic
    hist = tedm_synthetic.run(steps=40, size=64, n_train=16, n_val=8, batch=8, lr=1e-3, verbose=False)
    loss0, dice0 = hist[0][0], hist[0][1]
    loss1, dice1 = hist[-1][0], hist[-1][1]
    print(f'val loss {loss0:.3f} -> {loss1:.3f}, held-out Dice {dice0:.3f} -> {dice1:.3f}')
    assert loss1 < loss0 and dice1 > 0.8
