"""`DatasetDM` -- drop-in for models/datasetDM_model.py:30-88 (LEDM / LEDMe / TEDM feature extractor).

The reference hooks `diffusion_model.model.ups[i][2]`, runs the UNet once per timestep, copies every
hooked tensor to the CPU, resizes there and moves the 960*T-channel result back to the device
(:16-27,67-88).  Here the T noised copies are stacked into ONE native forward of B*T images (the UNet
has no cross-sample op), the tail of the decoder that feeds no tap is skipped (TEDM_FWD_TAPS_ONLY),
and the nearest-resize + channel concat is a HIP kernel writing the (B, 960*T, S, S) tensor directly
on the device.  Values and channel order ([t0: tap0..tap3, t1: ...]) are the reference's.
"""
from __future__ import annotations

import os
from argparse import Namespace
from typing import Optional

import torch
from torch import Tensor, nn

from . import _lib
from .diffusion import DiffusionModel


class DatasetDM(nn.Module):
    def __init__(self, args: Namespace) -> None:
        super().__init__()
        path = getattr(args, 'saved_diffusion_model', None)
        if not path or not os.path.isfile(path):                              # datasetDM_model.py:34-37
            self.diffusion_model = DiffusionModel(args)
            if getattr(args, 'verbose', False):
                print(f'No model found at {path}. Please load model!')
        else:                                                                  # :38-43 (Namespace pickled inside)
            from .checkpoint import load_checkpoint
            ckpt = load_checkpoint(path, map_location=torch.device(getattr(args, 'device', 'cpu')),
                                   allow_unsafe=bool(getattr(args, 'allow_unsafe_checkpoint', False)))
            self.diffusion_model = DiffusionModel(ckpt['config'])
            self.diffusion_model.load_state_dict(ckpt['model_state_dict'])
        self.diffusion_model.eval()
        self._features = {}
        self.steps = list(args.t_steps_to_save)
        self.max_images_per_forward = int(getattr(args, 'max_images_per_forward', 64))
        feat_ch = sum(s[3] for s in self.diffusion_model.model._tap_shapes(1, 8, 8))   # 960 for the default UNet
        self.feature_channels = feat_ch
        self.fused_head = bool(getattr(args, 'fused_head', True))
        self.classifier = nn.Sequential(                                       # datasetDM_model.py:57-64
            nn.Conv2d(feat_ch * len(self.steps), 128, 1), nn.ReLU(), nn.BatchNorm2d(128),
            nn.Conv2d(128, 32, 1), nn.ReLU(), nn.BatchNorm2d(32), nn.Conv2d(32, 1, 1))

    @torch.no_grad()
    def extract_features(self, x_0: Tensor, noise: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
        """datasetDM_model.py:67-83.  Returns (B, feature_channels*T, S, S) fp32 on x_0's device
        (written into `out` when given: the tensor is 40 GB at B=32, T=5, 256x256)."""
        if noise is not None:
            assert x_0.shape == noise.shape
        dm, unet = self.diffusion_model, self.diffusion_model.model
        B, _, H, W = x_0.shape
        assert H == W, 'extract_features resizes to [x.shape[-1]]*2 (datasetDM_model.py:81): square inputs only'
        S, T, dev = W, len(self.steps), x_0.device
        shape = (B, self.feature_channels * T, S, S)
        if out is not None:
            assert tuple(out.shape) == shape and out.dtype == torch.float32 and out.is_contiguous() and out.device == dev
        feats = out if out is not None else torch.empty(shape, dtype=torch.float32, device=dev)
        L = _lib.lib()
        per_call = max(1, self.max_images_per_forward // B) if B <= self.max_images_per_forward else 1
        for s0 in range(0, T, per_call):
            steps = self.steps[s0:s0 + per_call]
            xs, ts = [], []
            for st in steps:
                t = torch.full((B,), int(st), dtype=torch.long, device=dev)
                x_t, _ = dm.forward_diffusion_model(x_0=x_0, t=t, noise=noise)   # un-normalised x_0 (:76)
                xs.append(x_t)
                ts.append(t)
            _, taps = unet.native_forward(torch.cat(xs, 0), torch.cat(ts, 0), taps_only=True)
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                for j in range(len(steps)):
                    c_off = (s0 + j) * self.feature_channels
                    for tap in taps:                                         # tap: (B*Tc, C, h, w) view of NHWC
                        nhwc = tap.permute(0, 2, 3, 1)[j * B:(j + 1) * B]
                        assert nhwc.is_contiguous()
                        _, h, w, Cc = nhwc.shape
                        _lib.check(L.tedm_tap_to_features(nhwc.data_ptr(), B, h, w, Cc, feats.data_ptr(),
                                                          feats.shape[1], c_off, S, stream), 'tedm_tap_to_features')
                        c_off += Cc
        return feats

    @torch.no_grad()
    def _taps_per_step(self, x_0: Tensor, noise: Optional[Tensor] = None):
        """The decoder taps of every timestep in `self.steps` (what extract_features resizes and concatenates),
        kept at their native resolution: list over t of 4 NCHW-shaped views of NHWC buffers."""
        dm, unet = self.diffusion_model, self.diffusion_model.model
        B = x_0.shape[0]
        dev = x_0.device
        per_call = max(1, self.max_images_per_forward // B) if B <= self.max_images_per_forward else 1
        out = []
        for s0 in range(0, len(self.steps), per_call):
            steps = self.steps[s0:s0 + per_call]
            xs, ts = [], []
            for st in steps:
                t = torch.full((B,), int(st), dtype=torch.long, device=dev)
                x_t, _ = dm.forward_diffusion_model(x_0=x_0, t=t, noise=noise)
                xs.append(x_t)
                ts.append(t)
            _, taps = unet.native_forward(torch.cat(xs, 0), torch.cat(ts, 0), taps_only=True)
            for j in range(len(steps)):
                out.append([tp[j * B:(j + 1) * B] for tp in taps])
        return out

    def forward(self, x: Tensor, noise: Optional[Tensor] = None) -> Tensor:
        """datasetDM_model.py:85-88.  With `fused_head` (default) the first classifier layer runs at the taps' native
        resolution (tedm_amd/head.py) and the (B, 960*T, S, S) feature tensor is never built; the result equals
        `classifier(extract_features(x))` up to fp32 summation order.  `noise` pins the q-sample noise (tests)."""
        conv = self.classifier[0]
        fused = (self.fused_head and isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.out_channels % 16 == 0
                 and conv.in_channels == self.feature_channels * len(self.steps))
        if not fused:
            return self.classifier(self.extract_features(x, noise=noise).to(x.device))
        from .head import fused_first_layer
        taps = self._taps_per_step(x, noise)
        unet = self.diffusion_model.model                       # the head trains at the UNet's training precision
        prec = unet.train_precision if (torch.is_grad_enabled() and self.training) else unet.precision
        h1 = fused_first_layer(conv, taps, x.shape[-1], _lib.PREC_BY_NAME[prec])
        return self.classifier[1:](h1)
