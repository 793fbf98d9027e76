"""CPU study (test infrastructure, not product; lives under tests/ because it imports the oracle): which MFMA operand formats keep eps under the 1e-3 gate?

Monkeypatches F.conv2d inside the oracle so that every convolution's operands are rounded the way a
candidate MFMA scheme would round them, and prints the eps rel-L2 against the untouched fp32 oracle
on the golden 64x64 inputs.  Run: python tests/precision_study.py [scheme ...]
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import unet_ref as O                     # noqa: E402
from tedm_amd.weights import synth_state_dict        # noqa: E402

_conv = F.conv2d
CALLS = []


def bf16(x): return x.to(torch.bfloat16).float()
def fp16(x): return x.half().float()


def minifloat(x, ebits, mbits, bias, vmax):
    """round-to-nearest-even to a sign/ebits/mbits minifloat with subnormals, saturating at vmax"""
    ax = x.abs().clamp(min=1e-45)
    e = torch.floor(torch.log2(ax)).clamp(min=1 - bias)
    step = torch.exp2(e - mbits)
    q = torch.round(x / step) * step
    return q.clamp(-vmax, vmax)


def q15e(x, block=8, bits=15):
    """Block integer storage along the channel axis (dim 1): `bits`-bit signed integers with one shared power-of-two scale per
    `block` consecutive channels of a pixel (the exponent travels in the stolen LSBs of the block's eight int16: 16 B per 8
    channels, self-contained).  scale = 2^(floor(log2 amax) + 2 - bits), round to nearest, saturating."""
    xm = x.movedim(1, -1)
    C = xm.shape[-1]
    xb = xm.reshape(*xm.shape[:-1], C // block, block)
    amax = xb.abs().amax(dim=-1, keepdim=True).clamp(min=2.0 ** -100)
    sc = torch.exp2(torch.floor(torch.log2(amax)) + 2 - bits)
    lim = 2.0 ** (bits - 1) - 1
    q = torch.round(xb / sc).clamp(-lim, lim) * sc
    return q.reshape(*xm.shape).movedim(-1, 1)


def bf8(x): return minifloat(x, 5, 2, 15, 57344.0)
def fp8(x): return minifloat(x, 4, 3, 7, 448.0)


def mx(x, dim, kind, block=32):
    """OCP-MX block quantisation along `dim` in blocks of `block` with an E8M0 shared scale."""
    ebits, mbits, bias, vmax, emax = {'fp6': (2, 3, 1, 7.5, 2), 'bf6': (3, 2, 3, 28.0, 4),
                                      'fp4': (2, 1, 1, 6.0, 2), 'fp8': (4, 3, 7, 448.0, 8),
                                      'bf8': (5, 2, 15, 57344.0, 15)}[kind]
    xm = x.movedim(dim, -1)
    shp = xm.shape
    C = shp[-1]
    pad = (-C) % block
    if pad:
        xm = F.pad(xm, (0, pad))
    xb = xm.reshape(*xm.shape[:-1], -1, block)
    amax = xb.abs().amax(dim=-1, keepdim=True).clamp(min=2.0 ** -126)
    sc = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    q = minifloat(xb / sc, ebits, mbits, bias, vmax) * sc
    q = q.reshape(*xm.shape[:-1], -1)[..., :C]
    return q.movedim(-1, dim)


def mx4(x, dim, mode='ocp', lo_of=None):
    """fp4 e2m1 block quantiser as the HIP staging code does it.  mode 'ocp': scale 2^(floor(log2 amax)-2), saturating;
    'noclip': scale 2^ceil(log2(amax/6)).  lo_of: derive the scale from the block max of THAT tensor (the fp16-rounded
    operand) as 2^(max(E-11,-25)-2) instead of from x's own max."""
    xm = x.movedim(dim, -1)
    C = xm.shape[-1]
    pad = (-C) % 32
    if pad:
        xm = F.pad(xm, (0, pad))
    xb = xm.reshape(*xm.shape[:-1], -1, 32)
    if lo_of is not None:
        am = lo_of.movedim(dim, -1)
        if pad:
            am = F.pad(am, (0, pad))
        amax = am.reshape(*am.shape[:-1], -1, 32).abs().amax(dim=-1, keepdim=True).clamp(min=2.0 ** -126)
        E = torch.floor(torch.log2(amax))
        sc = torch.exp2(torch.clamp(E - 11, min=-25.0) - 2)
    else:
        amax = xb.abs().amax(dim=-1, keepdim=True).clamp(min=2.0 ** -126)
        if mode == 'ocp':
            sc = torch.exp2(torch.floor(torch.log2(amax)) - 2)
        else:
            sc = torch.exp2(torch.ceil(torch.log2(amax / 6.0)))
    q = minifloat(xb / sc, 2, 1, 1, 6.0) * sc
    q = q.reshape(*xm.shape[:-1], -1)[..., :C]
    return q.movedim(-1, dim)


def make_conv(scheme):
    def conv(x, w, b=None, stride=1, padding=0, *a, **k):
        if x.shape[1] == 1:                      # the stem runs on the exact-fp32 MFMA
            return _conv(x, w, b, stride, padding, *a, **k)
        c = lambda xx, ww: _conv(xx, ww, None, stride, padding, *a, **k)
        if scheme.startswith('la16'):
            # la16[q|o]: single fp16 product on the linear-attention projections (to_qkv: 384 outputs; to_out: 128 inputs, 1x1),
            # the product set of the k x k convs (ResnetBlock-internal below full resolution) otherwise
            is_q = w.shape[2] == 1 and w.shape[0] == 384
            is_o = w.shape[2] == 1 and w.shape[1] == 128 and w.shape[0] != 384
            if (is_q and not scheme.endswith('o')) or (is_o and not scheme.endswith('q')):
                y = c(fp16(x), fp16(w))
                return y if b is None else y + b.reshape(1, -1, 1, 1)
            return make_conv('set:5-8,10-13,15-18,20-27,29-32,34-37')(x, w, b, stride, padding, *a, **k)
        if scheme.startswith('set:'):
            # set:<i,j,k-l,...>: single fp16 product on those k x k convolutions (call order), f16f4 elsewhere
            if w.shape[2] > 1:
                CALLS.append((x.shape[1], w.shape[0], x.shape[-1], w.shape[2]))
                idx, sel = len(CALLS) - 1, False
                for part in scheme[4:].split(','):
                    lo, _, hi = part.partition('-')
                    sel |= int(lo) <= idx <= int(hi or lo)
                if sel:
                    y = c(fp16(x), fp16(w))
                    return y if b is None else y + b.reshape(1, -1, 1, 1)
            return make_conv('f16f4')(x, w, b, stride, padding, *a, **k)
        if scheme.startswith('idx'):
            # idx<N>: single fp16 product on the N-th k x k convolution in call order only (CALLS counts them), f16f4 elsewhere
            if w.shape[2] > 1:
                CALLS.append((x.shape[1], w.shape[0], x.shape[-1], w.shape[2]))
                if len(CALLS) - 1 == int(scheme[3:]):
                    y = c(fp16(x), fp16(w))
                    return y if b is None else y + b.reshape(1, -1, 1, 1)
            return make_conv('f16f4')(x, w, b, stride, padding, *a, **k)
        if scheme.startswith('only'):
            # only<S>[c1|c2]: single fp16 product on the k x k convs whose OUTPUT is S pixels wide (one resolution level of the
            # 64x64 study input: 64, 32, 16, 8), f16f4 elsewhere; c1 / c2: only convs WITHOUT / WITH a GroupNorm'd input
            # (told apart by the input statistics is not possible here: c1 = Cin != Cout or first-in-block is approximated by
            # "input is not post-SiLU", i.e. its minimum is below -0.3)
            S_ = int(''.join(ch for ch in scheme[4:] if ch.isdigit()))
            sel = w.shape[2] > 1 and (x.shape[-1] // (stride if isinstance(stride, int) else stride[0])) == S_
            if sel and scheme.endswith('c1'):
                sel = float(x.min()) < -0.3
            if sel and scheme.endswith('c2'):
                sel = float(x.min()) >= -0.3
            if sel:
                y = c(fp16(x), fp16(w))
                return y if b is None else y + b.reshape(1, -1, 1, 1)
            return make_conv('f16f4')(x, w, b, stride, padding, *a, **k)
        if scheme.startswith('mixed'):
            # mixed<N>[h]: single fp16 product for convs with >= N output channels, f16f4 below
            # ('h': fp16 x (fp16 hi + lo) weights, i.e. only the ACTIVATION is rounded once)
            thr = int(''.join(ch for ch in scheme[5:] if ch.isdigit()))
            if w.shape[0] >= thr and w.shape[2] > 1:
                if scheme.endswith('h'):
                    y = c(fp16(x), w)
                else:
                    y = c(fp16(x), fp16(w))
                return y if b is None else y + b.reshape(1, -1, 1, 1)
            return make_conv('f16f4')(x, w, b, stride, padding, *a, **k)
        if scheme == 'bf16':
            y = c(bf16(x), bf16(w))
        elif scheme == 'fp16':
            y = c(fp16(x), fp16(w))
        elif scheme == 'fp16_a_only':
            y = c(fp16(x), w)
        elif scheme == 'bf16x3':
            xh, wh = bf16(x), bf16(w)
            xl, wl = bf16(x - xh), bf16(w - wh)
            y = c(xh, wh) + c(xl, wh) + c(xh, wl)
        elif scheme.startswith('fp16+'):
            kind = scheme.split('+')[1]           # bf8 / fp8 (unscaled, 2^12 prescale) or mx kinds
            xh, wh = fp16(x), fp16(w)
            xl, wl = x - xh, w - wh
            if kind in ('bf8', 'fp8'):
                q = bf8 if kind == 'bf8' else fp8
                s = 2.0 ** 12
                y = c(xh, wh) + c(q(xl * s) / s, q(w)) + c(q(x), q(wl * s) / s)
            else:
                kind = kind[2:]                   # mxfp6 -> fp6 ...
                y = c(xh, wh) + c(mx(xl, 1, kind), mx(w, 1, kind)) + c(mx(x, 1, kind), mx(wl, 1, kind))
        elif scheme.startswith('f16f4'):
            # f16f4[-noclip][-dlo]: activations and weights fp16 hi + fp4 lo / fp4 copy
            mode = 'noclip' if 'noclip' in scheme else 'ocp'
            xc = x.clamp(-65504.0, 65504.0)
            xh, wh = fp16(xc), fp16(w)
            xl, wl = x - xh, w - wh
            xl4 = mx4(xl, 1, mode, lo_of=xh if 'dlo' in scheme else None)
            y = c(xh, wh) + c(xl4, mx4(w, 1, mode)) + c(mx4(x, 1, mode), mx4(wl, 1, mode))
        elif scheme.startswith('bf16+'):
            kind = scheme.split('+')[1][2:]
            xh, wh = bf16(x), bf16(w)
            xl, wl = x - xh, w - wh
            y = c(xh, wh) + c(mx(xl, 1, kind), mx(w, 1, kind)) + c(mx(x, 1, kind), mx(wl, 1, kind))
        else:
            raise ValueError(scheme)
        return y if b is None else y + b.reshape(1, -1, 1, 1)
    return conv


def main():
    gd = ROOT / 'tests' / 'golden'
    size = 64
    if sys.argv[1:] and sys.argv[1].isdigit():
        size = int(sys.argv.pop(1))
    g = np.load(gd / f'eps_d64_s{size}.npz')
    shapes = {k: tuple(v) for k, v in json.loads((gd / 'unet_d64_state_dict_keys.json').read_text()).items()}
    w = synth_state_dict(shapes, 0)
    x_t, t = torch.from_numpy(g['x_t']), torch.from_numpy(g['t'])
    ref = O.unet_forward(w, x_t, t)
    schemes = sys.argv[1:] or ['bf16', 'fp16', 'fp16_a_only', 'bf16x3', 'fp16+bf8', 'fp16+fp8', 'fp16+mxfp8', 'fp16+mxbf8',
                               'fp16+mxfp6', 'fp16+mxbf6', 'fp16+mxfp4', 'bf16+mxfp8', 'bf16+mxfp6']
    for s in schemes:
        F.conv2d = make_conv(s)
        try:
            eps = O.unet_forward(w, x_t, t)
        finally:
            F.conv2d = _conv
        extra = ''
        if s.startswith('idx') and int(s[3:]) < len(CALLS):
            extra = '  conv (Cin, Cout, width, k) = %s of %d' % (CALLS[int(s[3:])], len(CALLS))
        CALLS.clear()
        print(f'{s:14s} eps rel-L2 = {float((eps - ref).norm() / ref.norm()):.3e}{extra}', flush=True)




def storage_study(size=64):
    """What would 16-bit activation storage cost?  Every tensor the HIP path writes to HBM -- conv outputs, resnet-block
    outputs, attention-block outputs -- is rounded to the storage format in the fp32 oracle (contractions: f16f4).
    Run: python tests/precision_study.py storage [64|256]"""
    gd = ROOT / 'tests' / 'golden'
    g = np.load(gd / f'eps_d64_s{size}.npz')
    shapes = {k: tuple(v) for k, v in json.loads((gd / 'unet_d64_state_dict_keys.json').read_text()).items()}
    w = synth_state_dict(shapes, 0)
    x_t, t = torch.from_numpy(g['x_t']), torch.from_numpy(g['t'])
    ref = O.unet_forward(w, x_t, t)
    res0, la0, fa0 = O.resnet_block, O.linear_attention, O.full_attention
    for arith in ('fp32', 'f16f4'):
        for name, rnd in (('fp32', lambda v: v), ('fp16', fp16), ('bf16', bf16), ('q15e8', q15e), ('q16b32', lambda v: q15e(v, 32, 16))):
            base = make_conv(arith) if arith != 'fp32' else _conv

            def conv(x, ww, b=None, *a, _base=base, _rnd=rnd, **k):
                y = _base(x, ww, b, *a, **k)
                return y if y.shape[1] == 1 else _rnd(y)            # the 1-channel final conv is the fp32 result
            F.conv2d = conv
            O.resnet_block = lambda *a, _r=rnd, **k: _r(res0(*a, **k))
            O.linear_attention = lambda *a, _r=rnd, **k: _r(la0(*a, **k))
            O.full_attention = lambda *a, _r=rnd, **k: _r(fa0(*a, **k))
            try:
                eps = O.unet_forward(w, x_t, t)
            finally:
                F.conv2d = _conv
                O.resnet_block, O.linear_attention, O.full_attention = res0, la0, fa0
            print(f'contractions {arith:6s} storage {name:5s} eps rel-L2 = {float((eps - ref).norm() / ref.norm()):.3e}', flush=True)


WHICH = None      # 'block1' / 'block2': round only that conv's output


def branch_storage_study(size=64):
    """16-bit storage of the BRANCH tensors only: the two conv outputs inside every ResnetBlock (consumed once, by the
    GroupNorm prologue of the next kernel) -- the residual stream (block and attention outputs) stays fp32.
    Run: python tests/precision_study.py branch [64|256]"""
    gd = ROOT / 'tests' / 'golden'
    g = np.load(gd / f'eps_d64_s{size}.npz')
    shapes = {k: tuple(v) for k, v in json.loads((gd / 'unet_d64_state_dict_keys.json').read_text()).items()}
    w = synth_state_dict(shapes, 0)
    x_t, t = torch.from_numpy(g['x_t']), torch.from_numpy(g['t'])
    ref = O.unet_forward(w, x_t, t)
    block0 = O.block
    for arith in ('fp32', 'f16f4'):
        for name, rnd in (('fp32', lambda v: v), ('fp16', fp16), ('q15e8', q15e), ('q16b32', lambda v: q15e(v, 32, 16))):
            def blk(p, pre, x, groups, scale_shift, _rnd=rnd):
                y = F.conv2d(x, p[pre + '.proj.weight'], p[pre + '.proj.bias'], padding=1)
                if WHICH is None or pre.endswith(WHICH):
                    y = _rnd(y)
                y = F.group_norm(y, groups, p[pre + '.norm.weight'], p[pre + '.norm.bias'], eps=1e-5)
                if scale_shift is not None:
                    y = y * (scale_shift[0] + 1) + scale_shift[1]
                return F.silu(y)
            F.conv2d = make_conv(arith) if arith != 'fp32' else _conv
            O.block = blk
            try:
                eps = O.unet_forward(w, x_t, t)
            finally:
                F.conv2d = _conv
                O.block = block0
            print(f'contractions {arith:6s} branch storage {name:5s} eps rel-L2 = {float((eps - ref).norm() / ref.norm()):.3e}', flush=True)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'branch':
    if len(sys.argv) > 3:
        WHICH = sys.argv[3]
    branch_storage_study(int(sys.argv[2]) if len(sys.argv) > 2 else 64)
    sys.exit(0)
if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'storage':
    storage_study(int(sys.argv[2]) if len(sys.argv) > 2 else 64)
    sys.exit(0)
elif __name__ == '__main__':
    main()
