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
        self.fused_tail = bool(getattr(args, 'fused_tail', os.environ.get('TEDM_FUSED_TAIL', '1') != '0'))     # the rest of the head as HIP kernels: one under no_grad (head.fused_head_eval), six passes in training (head.FusedTailTrain)
        # TEDM (trainers/train_datasetDM.py:30-42): ONE classifier shared by all timesteps.  The reference swaps it in after
        # construction; both routes work here (`args.shared_weights_over_timesteps`, config.py, or assigning
        # `model.classifier = nn.Sequential(Rearrange(...), nn.Conv2d(960, 128, 1), ...)` later).
        self.shared_weights_over_timesteps = bool(getattr(args, 'shared_weights_over_timesteps', False))
        if self.shared_weights_over_timesteps:
            from .tedm import shared_weights_classifier
            self.classifier = shared_weights_classifier(len(self.steps), feat_ch, int(getattr(args, 'out_channels', 1)))
        else:
            self.classifier = nn.Sequential(                                   # datasetDM_model.py:57-64
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

    @torch.no_grad()
    def _taps_steps_in_batch(self, x_0: Tensor, noise: Optional[Tensor] = None):
        """The decoder taps of the B*T noised images in '(b step)' order (row b*T + step: the order the shared-weights
        classifier's Rearrange produces, trainers/train_datasetDM.py:34) as 4 NCHW-shaped views of NHWC buffers
        (B*T, C_i, h_i, w_i).  The UNet has no cross-sample op, so stacking the timesteps into the batch is exact; the
        native forwards are cut at `max_images_per_forward` images along b."""
        dm, unet = self.diffusion_model, self.diffusion_model.model
        B, T, dev = x_0.shape[0], len(self.steps), x_0.device
        steps = torch.tensor([int(s) for s in self.steps], dtype=torch.long, device=dev)
        nb = max(1, self.max_images_per_forward // T)
        chunks = []
        for b0 in range(0, B, nb):
            xb = x_0[b0:b0 + nb]
            n = xb.shape[0]
            xr = xb.repeat_interleave(T, dim=0)
            t = steps.repeat(n)
            nz = noise[b0:b0 + nb].repeat_interleave(T, dim=0) if noise is not None else None   # one noise per image (:68-76)
            x_t, _ = dm.forward_diffusion_model(x_0=xr, t=t, noise=nz)                           # un-normalised x_0 (:76)
            _, taps = unet.native_forward(x_t, t, taps_only=True)
            chunks.append(taps)
        if len(chunks) == 1:
            return chunks[0]
        return [torch.cat([c[i].permute(0, 2, 3, 1) for c in chunks], 0).permute(0, 3, 1, 2) for i in range(len(chunks[0]))]

    def head_layout(self):
        """('concat', i) : classifier[i] is the reference's Conv2d(960*T, ., 1) on the channel concat (datasetDM_model.py:57-64);
        ('shared', i)   : classifier[0] regroups '(step act) -> (b step)' and classifier[i] is Conv2d(960, ., 1) (TEDM);
        None            : anything else -> the unfused path `classifier(extract_features(x))`."""
        from .tedm import is_steps_to_batch
        c = self.classifier
        if not isinstance(c, nn.Sequential) or len(c) == 0:
            return None

        def ok(conv, cin):
            return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1
                    and conv.out_channels % 16 == 0 and conv.in_channels == cin)
        if is_steps_to_batch(c[0]):
            n = getattr(c[0], 'n_steps', None) or dict(getattr(c[0], 'axes_lengths', ())).get('step')
            if n == len(self.steps) and len(c) > 1 and ok(c[1], self.feature_channels):
                return ('shared', 1)
            return None
        if ok(c[0], self.feature_channels * len(self.steps)):
            return ('concat', 0)
        return None

    def forward(self, x: Tensor, noise: Optional[Tensor] = None) -> Tensor:
        """datasetDM_model.py:85-88.  With `fused_head` (default) the first classifier layer runs at the taps' native
        resolution (tedm_amd/head.py) and the (B, 960*T, S, S) feature tensor is never built; the result equals
        `classifier(extract_features(x))` up to fp32 summation order.  `noise` pins the q-sample noise (tests).
        Shared-weights (TEDM) head: output (B*T, 1, S, S) in '(b step)' order; the 960 -> 128 projection runs ONCE per tap
        over the B*T images with the same weights (no accumulation over t)."""
        layout = self.head_layout() if self.fused_head else None
        if layout is None:
            return self.classifier(self.extract_features(x, noise=noise).to(x.device))
        from .head import eval_tail_supported, fused_first_layer, fused_head_eval, fused_tail_train, train_tail_supported
        kind, ci = layout
        unet = self.diffusion_model.model                       # the head trains at the UNet's training precision
        prec = unet.train_precision if (torch.is_grad_enabled() and self.training) else unet.precision
        taps = [self._taps_steps_in_batch(x, noise)] if kind == 'shared' else self._taps_per_step(x, noise)
        conv, tail = self.classifier[ci], list(self.classifier[ci + 1:])
        if self.fused_tail and not torch.is_grad_enabled() and eval_tail_supported(tail, conv.out_channels):
            # inference: up-sampling sum, ReLU, both folded BatchNorms and the 128 -> 32 -> 1 layers in ONE kernel after the projections
            return fused_head_eval(conv, tail, taps, x.shape[-1], _lib.PREC_BY_NAME[prec])
        h1 = fused_first_layer(conv, taps, x.shape[-1], _lib.PREC_BY_NAME[prec])
        if self.fused_tail and self.training and torch.is_grad_enabled() and train_tail_supported(tail, conv.out_channels):
            # training: the tail's ~15 torch tensor passes as three forward and three backward kernels (csrc/head_train.hip)
            return fused_tail_train(tail, h1)
        return self.classifier[ci + 1:](h1)
