"""Pixel-classifier head with the first layer applied at the taps' native resolution.

Reference: models/datasetDM_model.py:57-64,85-88 -- `classifier(extract_features(x))`, where extract_features
materialises the (B, 960*T, S, S) concat of nearest-upsampled decoder taps (40 GB at B=32, T=5, S=256) and the
classifier starts with Conv2d(960*T, 128, 1).  A 1x1 conv commutes with nearest upsampling and distributes over the
channel concat,  W1 * cat_i(up(f_i)) = sum_i up(W1_i * f_i),  so this module

  forward : z_i (+)= tedm_conv1x1_nhwc(tap_{t,i}, W1[:, slice(t,i)])   at (h_i, w_i)   for every timestep t
            h1 = b1 + sum_i nearest_up(z_i)                           (tedm_upsample_sum_nhwc)
  backward: dz_i = block sums of dh1 (tedm_pool_sum_nhwc);  dW1[:, slice(t,i)] = dz_i^T tap_{t,i} (tedm_conv1x1_wgrad_nhwc)

and never builds the feature tensor.  Results equal the reference's up to fp32 summation order (the projection is
computed with the bf16x3 matrix-core scheme of the UNet path, ~2e-5 relative).  The rest of the head
(ReLU/BatchNorm/1x1 convs on 128 and 32 channels) stays the module's own torch layers.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch
from torch import Tensor

from . import _lib


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def project_taps(W1: Tensor, prec: int, n_taps: int, taps: Sequence[Tensor]) -> List[Tensor]:
    """z_i = sum_t tedm_conv1x1_nhwc(tap_{t,i}, W1[:, slice(t, i)]) at every tap's native resolution (accumulated over the
    timesteps through the kernel's epilogue addend).  taps: flat [t0: tap0..tap{n-1}, t1: ...] NHWC fp32; returns n_taps tensors
    (B, h_i, w_i, Cout)."""
    L = _lib.lib()
    dev = W1.device
    Cout = W1.shape[0]
    T = len(taps) // n_taps
    assert T * n_taps == len(taps) and T >= 1
    B = taps[0].shape[0]
    w2d = W1.detach().reshape(Cout, -1).float()
    assert w2d.shape[1] == sum(t.shape[3] for t in taps), 'classifier[0].in_channels does not match the taps'
    st = _stream(dev)
    z: List[Tensor] = []
    off = 0
    for k, tap in enumerate(taps):
        i = k % n_taps
        assert tap.is_contiguous() and tap.dtype == torch.float32 and tap.shape[0] == B
        _, h, w, Ci = tap.shape
        nbytes = L.tedm_conv1x1_packed_bytes(Cout, Ci)
        if nbytes == 0:
            raise _lib.TedmError(f'fused head needs Cout % 16 == 0 and tap channels % 32 == 0 (got {Cout}, {Ci})')
        wslice = w2d[:, off:off + Ci].contiguous()          # stream-ordered allocations: safe to drop after the enqueue
        wpk = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(L.tedm_conv1x1_pack(wslice.data_ptr(), Cout, Ci, wpk.data_ptr(), prec, st), 'tedm_conv1x1_pack')
        if k < n_taps:
            z.append(torch.empty(B, h, w, Cout, dtype=torch.float32, device=dev))
        _lib.check(L.tedm_conv1x1_nhwc(tap.data_ptr(), B, h, w, Ci, wpk.data_ptr(), Cout, None,
                                       z[i].data_ptr() if k >= n_taps else None, z[i].data_ptr(), prec, st),
                   'tedm_conv1x1_nhwc')
        off += Ci
    return z


class FusedFirstLayer(torch.autograd.Function):
    """h1 = Conv2d(960*T, Cout, 1)(cat_t cat_i up(tap_{t,i}))  without the concat.

    apply(W1 (Cout, sum_t sum_i C_i, 1, 1), b1 (Cout,) or None, S, prec, n_taps, *taps) where taps is the flat list
    [t0: tap0..tap{n-1}, t1: ...] of NHWC-contiguous (B, h_i, w_i, C_i) fp32 CUDA tensors.  Returns (B, Cout, S, S) as a
    channels_last view."""

    @staticmethod
    def forward(ctx, W1: Tensor, b1, S: int, prec: int, n_taps: int, *taps: Tensor):
        L = _lib.lib()
        dev = W1.device
        Cout = W1.shape[0]
        B = taps[0].shape[0]
        with torch.cuda.device(dev):
            st = _stream(dev)
            z = project_taps(W1, prec, n_taps, taps)
            out = torch.empty(B, S, S, Cout, dtype=torch.float32, device=dev)
            zp = (C.c_void_p * n_taps)(*[t.data_ptr() for t in z])
            hh = (C.c_int * n_taps)(*[t.shape[1] for t in z])
            ww = (C.c_int * n_taps)(*[t.shape[2] for t in z])
            bias = b1.detach().float().contiguous() if b1 is not None else None
            _lib.check(L.tedm_upsample_sum_nhwc(zp, hh, ww, n_taps, B, S, Cout, bias.data_ptr() if bias is not None else None,
                                                out.data_ptr(), st), 'tedm_upsample_sum_nhwc')
        ctx.save_for_backward(*taps)
        ctx.meta = (S, n_taps, Cout, b1 is not None, tuple(W1.shape))
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dh1: Tensor):
        L = _lib.lib()
        taps = ctx.saved_tensors
        S, n_taps, Cout, has_bias, wshape = ctx.meta
        dev = dh1.device
        g = dh1.permute(0, 2, 3, 1).contiguous().float()            # (B, S, S, Cout) NHWC
        B = g.shape[0]
        dW = torch.empty(Cout, sum(t.shape[3] for t in taps), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _stream(dev)
            # block sums of dh1 per tap resolution, as a cascade: a coarser level pools the next finer one that divides it (one read
            # of the full-resolution gradient instead of one per level)
            pooled = {1: g}
            for i in range(n_taps):
                _, h, w, _ = taps[i].shape
                assert S % h == 0 and S % w == 0 and S // h == S // w, 'taps must be integer down-scalings of the image'
            for f in sorted({S // taps[i].shape[1] for i in range(n_taps)}):
                if f in pooled:
                    continue
                base = max(k for k in pooled if f % k == 0)
                src = pooled[base]
                d = torch.empty(B, S // f, S // f, Cout, dtype=torch.float32, device=dev)
                _lib.check(L.tedm_pool_sum_nhwc(src.data_ptr(), B, S // f, S // f, f // base, Cout, d.data_ptr(), st), 'tedm_pool_sum_nhwc')
                pooled[f] = d
            dz = [pooled[S // taps[i].shape[1]] for i in range(n_taps)]
            off = 0
            for k, tap in enumerate(taps):
                i = k % n_taps
                _, h, w, Ci = tap.shape
                need = L.tedm_conv1x1_wgrad_workspace_bytes(Cout, Ci, B, h, w)
                ws = torch.empty(need, dtype=torch.uint8, device=dev)
                dwk = torch.empty(Cout, Ci, dtype=torch.float32, device=dev)
                _lib.check(L.tedm_conv1x1_wgrad_nhwc(dz[i].data_ptr(), tap.data_ptr(), B, h, w, Ci, Cout, dwk.data_ptr(), 0,
                                                     ws.data_ptr(), need, st), 'tedm_conv1x1_wgrad_nhwc')
                dW[:, off:off + Ci] = dwk
                off += Ci
        # the bias gradient is the sum over ALL pixels: taken from the coarsest block sums (up to 64x fewer elements than dh1)
        db = pooled[max(pooled)].sum(dim=(0, 1, 2)) if has_bias else None
        return (dW.view(wshape), db, None, None, None) + (None,) * len(taps)


def fused_first_layer(conv: torch.nn.Conv2d, taps_by_step: Sequence[Sequence[Tensor]], S: int, prec: int = 0) -> Tensor:
    """taps_by_step[t][i]: NCHW-shaped views (B, C_i, h_i, w_i) of NHWC buffers, as Unet.native_forward returns them."""
    return FusedFirstLayer.apply(conv.weight, conv.bias, S, prec, len(taps_by_step[0]), *_flat_nhwc(taps_by_step))


def _flat_nhwc(taps_by_step: Sequence[Sequence[Tensor]]) -> List[Tensor]:
    flat = []
    for taps in taps_by_step:
        for tp in taps:
            nhwc = tp.permute(0, 2, 3, 1)
            flat.append(nhwc if nhwc.is_contiguous() else nhwc.contiguous())
    return flat


def eval_tail_supported(tail: Sequence[torch.nn.Module], c1: int) -> bool:
    """True when `tail` (the classifier behind its first conv) is exactly the reference's ReLU, BatchNorm2d(c1), Conv2d(c1, 32, 1),
    ReLU, BatchNorm2d(32), Conv2d(32, 1, 1) (models/datasetDM_model.py:59-64) in eval mode: then it folds into one kernel."""
    nn = torch.nn
    if len(tail) != 6 or c1 % 32:
        return False
    r1, bn1, c2, r2, bn2, c3 = tail
    ok = (isinstance(r1, nn.ReLU) and isinstance(r2, nn.ReLU) and isinstance(bn1, nn.BatchNorm2d) and isinstance(bn2, nn.BatchNorm2d)
          and isinstance(c2, nn.Conv2d) and isinstance(c3, nn.Conv2d))
    if not ok:
        return False
    return (bn1.num_features == c1 and bn2.num_features == 32 and not bn1.training and not bn2.training
            and bn1.running_mean is not None and bn2.running_mean is not None
            and c2.kernel_size == (1, 1) and c2.in_channels == c1 and c2.out_channels == 32 and c2.groups == 1
            and c3.kernel_size == (1, 1) and c3.in_channels == 32 and c3.out_channels == 1)


@torch.no_grad()
def fused_head_eval(conv: torch.nn.Conv2d, tail: Sequence[torch.nn.Module], taps_by_step: Sequence[Sequence[Tensor]], S: int,
                    prec: int = 0) -> Tensor:
    """Inference: the whole classifier without its two wide intermediates.  First layer at the taps' native resolution
    (`project_taps`), then ONE kernel (`tedm_head_tail_nhwc`) that up-samples and sums the projections, applies ReLU, the
    128 -> 32 and 32 -> 1 layers with both eval-mode BatchNorms folded into the following conv.  Returns logits (B, 1, S, S)."""
    L = _lib.lib()
    _, bn1, c2, _, bn2, c3 = tail
    dev = conv.weight.device
    flat = _flat_nhwc(taps_by_step)
    n_taps = len(taps_by_step[0])
    with torch.cuda.device(dev):
        st = _stream(dev)
        z = project_taps(conv.weight, prec, n_taps, flat)
        c1 = conv.out_channels

        def fold(bn):
            sc = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps) if bn.affine else torch.rsqrt(bn.running_var.float() + bn.eps)
            sh = (bn.bias.float() if bn.affine else 0.0) - bn.running_mean.float() * sc
            return sc, sh
        s1, t1 = fold(bn1)
        s2, t2 = fold(bn2)
        w2 = c2.weight.float().reshape(32, c1)
        w2t = (w2 * s1[None, :]).t().contiguous()                                    # [C1][32]
        b2 = ((c2.bias.float() if c2.bias is not None else 0.0) + w2 @ t1).contiguous()
        w3 = c3.weight.float().reshape(32)
        w3f = (w3 * s2).contiguous()
        b3 = ((c3.bias.float() if c3.bias is not None else torch.zeros(1, device=dev)) + (w3 * t2).sum()).reshape(1).contiguous()
        b1 = (conv.bias.float() if conv.bias is not None else torch.zeros(c1, device=dev)).contiguous()
        B = z[0].shape[0]
        logits = torch.empty(B, 1, S, S, dtype=torch.float32, device=dev)
        zp = (C.c_void_p * n_taps)(*[t.data_ptr() for t in z])
        hh = (C.c_int * n_taps)(*[t.shape[1] for t in z])
        ww = (C.c_int * n_taps)(*[t.shape[2] for t in z])
        _lib.check(L.tedm_head_tail_nhwc(zp, hh, ww, n_taps, B, S, c1, b1.data_ptr(), w2t.data_ptr(), b2.data_ptr(), w3f.data_ptr(),
                                         b3.data_ptr(), logits.data_ptr(), st), 'tedm_head_tail_nhwc')
    return logits


# ---------------------------------------------------------------------------------------------------------------------
# Training-mode tail (ReLU, BatchNorm2d, Conv2d(c1, 32, 1), ReLU, BatchNorm2d, Conv2d(32, 1, 1)) in three forward and three
# backward passes over h1 (tedm_amd/csrc/head_train.hip).  torch runs it as ~15 tensor passes.

def train_tail_supported(tail: Sequence[torch.nn.Module], c1: int) -> bool:
    """True when `tail` is exactly the reference's classifier tail (models/datasetDM_model.py:59-64) in TRAINING mode with affine,
    running-statistics BatchNorms and a numeric momentum, and c1 is a width the kernels are built for."""
    nn = torch.nn
    if len(tail) != 6 or c1 not in (128, 256):
        return False
    r1, bn1, c2, r2, bn2, c3 = tail
    ok = (isinstance(r1, nn.ReLU) and isinstance(r2, nn.ReLU) and isinstance(bn1, nn.BatchNorm2d) and isinstance(bn2, nn.BatchNorm2d)
          and isinstance(c2, nn.Conv2d) and isinstance(c3, nn.Conv2d))
    if not ok:
        return False
    for bn, n in ((bn1, c1), (bn2, 32)):
        if not (bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None and bn.num_features == n):
            return False
    return (c2.kernel_size == (1, 1) and c2.in_channels == c1 and c2.out_channels == 32 and c2.groups == 1 and c2.bias is not None
            and c3.kernel_size == (1, 1) and c3.in_channels == 32 and c3.out_channels == 1 and c3.bias is not None)


def _bn_coeffs(part: Tensor, P: int, gamma: Tensor, beta: Tensor, eps: float, shift: Tensor):
    """Per-block partials [blocks][C][2] = (sum, sumsq) of (x - shift) -> batch mean, biased variance, rstd and y = a x + b (fp64
    reduction).  The kernels sum around `shift` (a per-channel estimate of the mean from the first pixels): E[x^2] - mean^2 from
    fp32 sums of the raw values loses the variance of a channel with |mean| >> std; around the shift it does not."""
    s = part.double().sum(0)
    dm = s[:, 0] / P
    mean = shift.double() + dm
    var = (s[:, 1] / P - dm * dm).clamp_min(0.0)
    rstd = torch.rsqrt(var + eps)
    a = gamma.double() * rstd
    b = beta.double() - mean * a
    return mean, var, rstd, a.float().contiguous(), b.float().contiguous()


class FusedTailTrain(torch.autograd.Function):
    """logits (B, 1, S, S) = tail(h1) for h1 given as the NHWC-contiguous (B, S, S, C1) buffer; returns also the batch statistics
    (mean1, var1, mean2, var2: biased) for the caller's running-statistics update."""

    BLOCKS = 1024

    @staticmethod
    def forward(ctx, h1: Tensor, g1, be1, w2, bias2, g2, be2, w3, b3, eps1: float, eps2: float):
        L = _lib.lib()
        dev = h1.device
        assert h1.is_contiguous() and h1.dtype == torch.float32 and h1.dim() == 4
        B, S, S2, C1 = h1.shape
        P = B * S * S2
        nb = FusedTailTrain.BLOCKS
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _stream(dev)
            part1 = torch.empty(nb, C1, 2, **f32)
            head_px = h1.view(P, C1)[:min(P, 256)]                                  # a few pixels: per-channel shifts for the statistics
            r1h = torch.relu(head_px)
            shift1 = r1h.mean(0).contiguous()
            _lib.check(L.tedm_head_train_stats1(h1.data_ptr(), P, C1, shift1.data_ptr(), part1.data_ptr(), nb, st), 'tedm_head_train_stats1')
            mean1, var1, rstd1, a1, b1 = _bn_coeffs(part1, P, g1.detach(), be1.detach(), eps1, shift1)
            w2m = w2.detach().reshape(32, C1).float().contiguous()
            w2t = w2m.t().contiguous()
            z2 = torch.empty(P, 32, **f32)
            part2 = torch.empty(nb, 32, 2, **f32)
            bias2f = bias2.detach().float().contiguous()
            shift2 = torch.relu((r1h * a1 + b1) @ w2t + bias2f).mean(0).contiguous()
            _lib.check(L.tedm_head_train_mid(h1.data_ptr(), P, C1, a1.data_ptr(), b1.data_ptr(), w2t.data_ptr(),
                                             bias2f.data_ptr(), shift2.data_ptr(), z2.data_ptr(), part2.data_ptr(), nb, st),
                       'tedm_head_train_mid')
            mean2, var2, rstd2, a2, b2 = _bn_coeffs(part2, P, g2.detach(), be2.detach(), eps2, shift2)
            w3v = w3.detach().reshape(32).float().contiguous()
            logits = torch.empty(B, 1, S, S2, **f32)
            _lib.check(L.tedm_head_train_out(z2.data_ptr(), P, a2.data_ptr(), b2.data_ptr(), w3v.data_ptr(),
                                             b3.detach().float().contiguous().data_ptr(), logits.data_ptr(), st), 'tedm_head_train_out')
        ctx.save_for_backward(h1, z2, a1, b1, a2, b2, w2m, w3v, mean1, rstd1, mean2, rstd2)
        ctx.meta = (P, C1, tuple(w2.shape), tuple(w3.shape))
        ctx.mark_non_differentiable(mean1, var1, mean2, var2)
        return logits, mean1, var1, mean2, var2

    @staticmethod
    def backward(ctx, dlogits: Tensor, *_unused):
        L = _lib.lib()
        h1, z2, a1, b1, a2, b2, w2m, w3v, mean1, rstd1, mean2, rstd2 = ctx.saved_tensors
        P, C1, w2shape, w3shape = ctx.meta
        dev = h1.device
        nb = FusedTailTrain.BLOCKS
        f32 = dict(dtype=torch.float32, device=dev)
        dl = dlogits.reshape(-1).float().contiguous()
        with torch.cuda.device(dev):
            st = _stream(dev)
            pa = torch.empty(nb, 33, **f32)
            _lib.check(L.tedm_head_train_bwd1(z2.data_ptr(), dl.data_ptr(), P, pa.data_ptr(), nb, st), 'tedm_head_train_bwd1')
            A = pa.double().sum(0)
            A1, A0 = A[:32], A[32]
            a2d, b2d, w3d = a2.double(), b2.double(), w3v.double()
            db3 = A0.reshape(1)
            dW3 = a2d * A1 + b2d * A0
            S1 = w3d * A0                                         # dbeta2
            S2 = w3d * (A1 - mean2 * A0) * rstd2                  # dgamma2
            k = torch.stack((a2d * w3d, -a2d * S1 / P + a2d * rstd2 * mean2 * S2 / P, -a2d * rstd2 * S2 / P)).float().contiguous()
            pg = torch.empty(nb, 32 * C1 + 32, **f32)
            _lib.check(L.tedm_head_train_bwd2(h1.data_ptr(), z2.data_ptr(), dl.data_ptr(), P, C1, k.data_ptr(), pg.data_ptr(), nb, st),
                       'tedm_head_train_bwd2')
            G = pg.double().sum(0)
            db2 = G[32 * C1:]
            G = G[:32 * C1].reshape(32, C1)
            a1d, b1d, w2d = a1.double(), b1.double(), w2m.double()
            dW2 = a1d[None, :] * G + b1d[None, :] * db2[:, None]
            T1 = (w2d * db2[:, None]).sum(0)                                       # dbeta1
            T2 = (w2d * (G - mean1[None, :] * db2[:, None])).sum(0) * rstd1        # dgamma1
            m = torch.stack((a1d, -a1d * T1 / P + a1d * rstd1 * mean1 * T2 / P, -a1d * rstd1 * T2 / P)).float().contiguous()
            dh1 = torch.empty_like(h1)
            _lib.check(L.tedm_head_train_bwd3(h1.data_ptr(), z2.data_ptr(), dl.data_ptr(), P, C1, k.data_ptr(), w2m.data_ptr(), m.data_ptr(),
                                              dh1.data_ptr(), st), 'tedm_head_train_bwd3')
        return (dh1, T2.float(), T1.float(), dW2.float().reshape(w2shape), db2.float(), S2.float(), S1.float(),
                dW3.float().reshape(w3shape), db3.float(), None, None)


def fused_tail_train(tail: Sequence[torch.nn.Module], h1_nchw: Tensor) -> Tensor:
    """tail(h1) in training mode through FusedTailTrain, including the BatchNorm running-statistics update (momentum rule and
    unbiased running variance of nn.BatchNorm2d).  h1_nchw: the (B, C1, S, S) channels-last VIEW FusedFirstLayer returns."""
    _, bn1, c2, _, bn2, c3 = tail
    h1 = h1_nchw.permute(0, 2, 3, 1)
    if not h1.is_contiguous():
        h1 = h1.contiguous()
    logits, mean1, var1, mean2, var2 = FusedTailTrain.apply(h1, bn1.weight, bn1.bias, c2.weight, c2.bias, bn2.weight, bn2.bias,
                                                            c3.weight, c3.bias, bn1.eps, bn2.eps)
    P = h1.shape[0] * h1.shape[1] * h1.shape[2]
    with torch.no_grad():
        for bn, mean, var in ((bn1, mean1, var1), (bn2, mean2, var2)):
            mom = bn.momentum
            bn.running_mean.mul_(1 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
            bn.running_var.mul_(1 - mom).add_((var * (P / max(P - 1, 1))).to(bn.running_var.dtype), alpha=mom)
            bn.num_batches_tracked += 1
    return logits
