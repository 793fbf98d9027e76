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
        T = len(taps) // n_taps
        assert T * n_taps == len(taps) and T >= 1
        B = taps[0].shape[0]
        w2d = W1.detach().reshape(Cout, -1).float()
        assert w2d.shape[1] == sum(t.shape[3] for t in taps), 'classifier[0].in_channels does not match the taps'
        with torch.cuda.device(dev):
            st = _stream(dev)
            z: List[Tensor] = []
            off = 0
            keep = []                                   # packed weight buffers must outlive the enqueued kernels
            for k, tap in enumerate(taps):
                i = k % n_taps
                assert tap.is_contiguous() and tap.dtype == torch.float32 and tap.shape[0] == B
                _, h, w, Ci = tap.shape
                nbytes = L.tedm_conv1x1_packed_bytes(Cout, Ci)
                if nbytes == 0:
                    raise _lib.TedmError(f'fused head needs Cout % 16 == 0 and tap channels % 32 == 0 (got {Cout}, {Ci})')
                wslice = w2d[:, off:off + Ci].contiguous()
                wpk = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                keep += [wslice, wpk]
                _lib.check(L.tedm_conv1x1_pack(wslice.data_ptr(), Cout, Ci, wpk.data_ptr(), prec, st), 'tedm_conv1x1_pack')
                if k < n_taps:
                    z.append(torch.empty(B, h, w, Cout, dtype=torch.float32, device=dev))
                _lib.check(L.tedm_conv1x1_nhwc(tap.data_ptr(), B, h, w, Ci, wpk.data_ptr(), Cout, None,
                                               z[i].data_ptr() if k >= n_taps else None, z[i].data_ptr(), prec, st),
                           'tedm_conv1x1_nhwc')
                off += Ci
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
            dz = []
            for i in range(n_taps):
                _, h, w, _ = taps[i].shape
                if h == S and w == S:
                    dz.append(g)
                    continue
                assert S % h == 0 and S // h == S // w, 'taps must be integer down-scalings of the image'
                d = torch.empty(B, h, w, Cout, dtype=torch.float32, device=dev)
                _lib.check(L.tedm_pool_sum_nhwc(g.data_ptr(), B, h, w, S // h, Cout, d.data_ptr(), st), 'tedm_pool_sum_nhwc')
                dz.append(d)
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
        db = g.sum(dim=(0, 1, 2)) if has_bias else None
        return (dW.view(wshape), db, None, None, None) + (None,) * len(taps)


def fused_first_layer(conv: torch.nn.Conv2d, taps_by_step: Sequence[Sequence[Tensor]], S: int, prec: int = 0) -> Tensor:
    """taps_by_step[t][i]: NCHW-shaped views (B, C_i, h_i, w_i) of NHWC buffers, as Unet.native_forward returns them."""
    flat = []
    for taps in taps_by_step:
        for tp in taps:
            nhwc = tp.permute(0, 2, 3, 1)
            flat.append(nhwc if nhwc.is_contiguous() else nhwc.contiguous())
    return FusedFirstLayer.apply(conv.weight, conv.bias, S, prec, len(taps_by_step[0]), *flat)
