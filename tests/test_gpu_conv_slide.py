"""GPU (MI355X): the sliding-window 3x3 convolution of the full-resolution 64-channel layers (tedm_amd/csrc/conv_slide.hip;
reference models/unet_model.py:119-135 `Block`) against the tiled implicit GEMM it replaces -- BIT FOR BIT, output and
GroupNorm partial sums -- against fp64 and against the CPU emulation of the f16f4 operand encoding."""
import sys
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent))
from test_gpu_parity import F4_TOL, dev, lib, rel, rnd, run_conv  # noqa: E402,F401

pytestmark = pytest.mark.gpu
PREC_F16F4 = 3


def tune(lib, **kv):
    from tedm_amd import _lib
    for k, v in kv.items():
        _lib.check(lib.tedm_set_tuning(k.encode(), int(v)), f'tune {k}')


# C1, H, W, B, gn, stats, rows per workgroup (0 = automatic)
SLIDE_CASES = [
    (64, 16, 32, 1, 0, 1, 0),       # one band per row range, one strip: every border at once
    (64, 64, 64, 2, 0, 1, 0),       # automatic row split (small grid)
    (64, 64, 64, 2, 1, 1, 64),      # whole strips, GroupNorm prologue
    (64, 40, 96, 3, 1, 0, 8),       # H not a power of two, one band per workgroup, no statistics
    (64, 128, 64, 1, 0, 1, 32),
    (64, 256, 256, 2, 1, 1, 0),     # the benchmark geometry at a small batch
    (64, 128, 128, 9, 1, 1, 0),     # odd batch: the automatic strip length avoids a partial round of workgroups (sl_pick_rows)
    (64, 64, 256, 5, 0, 1, 0),
]


@pytest.mark.parametrize('case', SLIDE_CASES, ids=lambda c: 'c{}_{}x{}_b{}_gn{}_st{}_r{}'.format(*c))
def test_slide_equals_tiled_kernel(lib, dev, case):
    C1, H, W, B, gn, stats, rows = case
    x = rnd(B, C1, H, W, seed=1)
    w = rnd(64, C1, 3, 3, seed=3, scale=(1.0 / (C1 * 9)) ** 0.5)
    b = rnd(64, seed=4, scale=0.1)
    g = (1 + 0.1 * rnd(B, C1, seed=5), 0.1 * rnd(B, C1, seed=6)) if gn else None
    try:
        tune(lib, conv_slide=0)
        ref_out, ref_st = run_conv(lib, dev, x, None, w, b, 3, 1, gn=g, stats_mode=stats, prec=PREC_F16F4)
        tune(lib, conv_slide=1, conv_slide_rows=rows)
        out, st = run_conv(lib, dev, x, None, w, b, 3, 1, gn=g, stats_mode=stats, prec=PREC_F16F4)
    finally:
        tune(lib, conv_slide=1, conv_slide_rows=0)
    assert torch.isfinite(out).all()
    xin = x if not gn else F.silu(x * g[0][:, :, None, None] + g[1][:, :, None, None])
    ref64 = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
    assert rel(out, ref64) < F4_TOL, rel(out, ref64)
    bad = (out != ref_out)
    assert not bad.any(), f'{int(bad.sum())} of {bad.numel()} outputs differ from the tiled kernel, first at {bad.nonzero()[0].tolist()}, rel {rel(out, ref_out):.3e}'
    if stats:
        assert torch.equal(st, ref_st)


# H, W, B, stats, rows per workgroup
SLIDE128_CASES = [(16, 32, 1, 1, 0), (64, 64, 2, 1, 0), (40, 96, 2, 0, 8), (64, 64, 2, 1, 64), (256, 256, 1, 1, 128), (64, 128, 5, 1, 0)]


@pytest.mark.parametrize('case', SLIDE128_CASES, ids=lambda c: '{}x{}_b{}_st{}_r{}'.format(*c))
def test_slide128_concat_conv(lib, dev, case):
    """128 -> 64 over the decoder's channel concat (conv_slide128.hip): the reduction is split over the waves, so the result is the
    tiled kernel's up to summation order -- against fp64, against the tiled kernel (rounding level), GroupNorm sums against
    the output they summarise, and bit-identical whatever the row split or the batch composition."""
    H, W, B, stats, rows = case
    x1, x2 = rnd(B, 64, H, W, seed=1), rnd(B, 64, H, W, seed=2)
    w = rnd(64, 128, 3, 3, seed=3, scale=(1.0 / (128 * 9)) ** 0.5)
    b = rnd(64, seed=4, scale=0.1)
    try:
        tune(lib, conv_slide=0)
        ref_out, ref_st = run_conv(lib, dev, x1, x2, w, b, 3, 1, stats_mode=stats, prec=PREC_F16F4)
        tune(lib, conv_slide=1, conv_slide_rows=rows)
        out, st = run_conv(lib, dev, x1, x2, w, b, 3, 1, stats_mode=stats, prec=PREC_F16F4)
        tune(lib, conv_slide_rows=8)
        out8, st8 = run_conv(lib, dev, x1, x2, w, b, 3, 1, stats_mode=stats, prec=PREC_F16F4)
        tune(lib, conv_slide_rows=0)
        out1, _ = run_conv(lib, dev, x1[:1], x2[:1], w, b, 3, 1, stats_mode=stats, prec=PREC_F16F4)
    finally:
        tune(lib, conv_slide=1, conv_slide_rows=0)
    assert torch.isfinite(out).all()
    ref64 = F.conv2d(torch.cat((x1, x2), 1).double(), w.double(), b.double(), padding=1)
    assert rel(out, ref64) < F4_TOL, rel(out, ref64)
    assert rel(out, ref_out) < 2e-6, rel(out, ref_out)
    assert torch.equal(out, out8) and torch.equal(out[:1], out1)
    if stats:
        assert torch.equal(st, st8)
        assert rel(st[..., 0].sum(1), out.double().sum(dim=(2, 3))) < 1e-5 or float((st[..., 0].sum(1) - out.double().sum(dim=(2, 3))).abs().max()) < 1e-2
        assert rel(st[..., 1].sum(1), (out.double() ** 2).sum(dim=(2, 3))) < 1e-5
        assert rel(st, ref_st) < 1e-4 or float((st - ref_st).abs().max()) < 1e-2


def test_slide_label_and_fallbacks(lib, dev):
    """Shapes the sliding kernel does not take fall through to the tiled kernel (no error, same numbers as with the switch off)."""
    for (C1, H, W) in [(64, 20, 32), (64, 32, 40)]:       # H % 8 != 0, W % 32 != 0
        x = rnd(1, C1, H, W, seed=7)
        w = rnd(64, C1, 3, 3, seed=8, scale=0.05)
        b = rnd(64, seed=9, scale=0.1)
        try:
            tune(lib, conv_slide=0)
            o0, _ = run_conv(lib, dev, x, None, w, b, 3, 1, prec=PREC_F16F4)
        finally:
            tune(lib, conv_slide=1)
        o1, _ = run_conv(lib, dev, x, None, w, b, 3, 1, prec=PREC_F16F4)
        assert torch.equal(o0, o1)
