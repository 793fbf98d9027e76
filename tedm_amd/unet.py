"""`Unet` -- drop-in for the reference denoiser (models/unet_model.py:246-368) whose forward runs
as hand-written HIP kernels on MI355X through libtedm_hip.so.

What is kept from the reference surface:
  * constructor signature and defaults (unet_model.py:247-259), `**kwargs` swallowed;
  * `forward(x, timestep=None, cond=None)` -- `cond` accepted and ignored (:333-368);
  * the module tree / attribute names, so `state_dict()` has the reference's 276 keys and a
    reference checkpoint loads unchanged; parameters are ordinary nn.Parameters (torch default
    initialisation, as in the reference);
  * `model.ups[i]` unpacks as (block1, block2, attn, upsample) and forward hooks registered on
    `attn` receive the post-residual decoder-attention output (models/datasetDM_model.py:50-53).

The sub-modules are parameter containers: the arithmetic of the whole forward is one call into the
native library (no per-layer PyTorch ops, no CPU fallback).  Calling a sub-module on its own raises.

Training: with autograd enabled and parameters requiring grad, `forward` goes through `_UnetTrainFn`
(`tedm_unet_forward_train` records a tape in a workspace, `tedm_unet_backward` writes all parameter gradients into
`grad_flat`); `native_forward_train` / `native_backward` are the same two calls without autograd.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _lib


def _native_only(name: str):
    raise NotImplementedError(
        f'{name}: sub-modules of tedm_amd.Unet are parameter containers; the arithmetic runs inside '
        'Unet.forward (libtedm_hip.so). Only ResnetBlock and the Residual(PreNorm(attention)) blocks can be called '
        'on their own (forward only, through Unet.module_forward).')


def _standalone(mod: nn.Module, x: Tensor, x_extra=None, time_emb: Optional[Tensor] = None) -> Tensor:
    """Stand-alone call of a ResnetBlock / Residual attention block of a tedm_amd.Unet (models/global_local_cl.py style)."""
    owner = getattr(mod, '_owner', None)
    owner = owner[0] if owner else None
    if owner is None:
        _native_only(type(mod).__name__)
    if torch.is_grad_enabled() and (x.requires_grad or (time_emb is not None and time_emb.requires_grad)):
        raise NotImplementedError('stand-alone sub-module calls are forward-only (the result carries no grad_fn); '
                                  'train through Unet.forward')
    return owner.module_forward(mod._qualname, x, time_emb)


class _Container(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        _native_only(type(self).__name__)


class SinusoidalPosEmb(_Container):
    """models/unet_model.py:76-93.  Holds the frequency table, computed on the CPU exactly as the
    reference does (exp(arange(half) * -(ln 1e4 / (half-1))) in fp32) and bound to the kernel."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        half = dim // 2
        freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
        self.register_buffer('freqs', freqs.float(), persistent=False)


class LearnedSinusoidalPosEmb(_Container):
    """models/unet_model.py:96-114 (`learned_sinusoidal_cond=True`): cat(t, sin(2 pi t w), cos(2 pi t w)) with the learned
    frequency vector `weights` (dim / 2); the arithmetic runs in `temb_kernel`, the gradient in `learned_sinusoid_bwd_kernel`."""

    def __init__(self, dim: int):
        super().__init__()
        assert dim % 2 == 0
        self.dim = dim
        self.weights = nn.Parameter(torch.randn(dim // 2))


class LayerNorm(_Container):
    """models/unet_model.py:52-61 (gain only)."""

    def __init__(self, dim: int):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))


class Block(_Container):
    """models/unet_model.py:119-135."""

    def __init__(self, dim: int, dim_out: int, groups: int = 8):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim_out, 3, padding=1)
        self.norm = nn.GroupNorm(groups, dim_out)
        self.act = nn.SiLU()


class ResnetBlock(_Container):
    """models/unet_model.py:138-175."""

    def __init__(self, dim: int, dim_out: int, *, time_emb_dim: Optional[int] = None, groups: int = 8):
        super().__init__()
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, dim_out * 2)) \
            if time_emb_dim is not None else None
        self.block1 = Block(dim, dim_out, groups=groups)
        self.block2 = Block(dim_out, dim_out, groups=groups)
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()

    def forward(self, x: Tensor, time_emb: Optional[Tensor] = None) -> Tensor:
        """models/unet_model.py:165-175 as one native call (inference only); x may be the torch.cat the reference builds."""
        return _standalone(self, x, None, time_emb)


class LinearAttention(_Container):
    """models/unet_model.py:178-210."""

    def __init__(self, dim: int, heads: int = 4, dim_head: int = 32):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        hidden = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Sequential(nn.Conv2d(hidden, dim, 1), LayerNorm(dim))


class Attention(_Container):
    """models/unet_model.py:213-241."""

    def __init__(self, dim: int, heads: int = 4, dim_head: int = 32, scale: int = 16):
        super().__init__()
        self.scale = scale
        self.heads = heads
        hidden = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hidden, dim, 1)


class PreNorm(_Container):
    """models/unet_model.py:64-73."""

    def __init__(self, dim: int, fn: nn.Module):
        super().__init__()
        self.fn = fn
        self.norm = LayerNorm(dim)


class Residual(nn.Module):
    """models/unet_model.py:29-36.  In this build the module is the *tap point*: Unet.forward hands
    it the tensor the native forward produced for this block and it returns it unchanged, so that
    forward hooks (DatasetDM, models/datasetDM_model.py:50-53) observe the reference's value."""

    def __init__(self, fn: nn.Module):
        super().__init__()
        self.fn = fn

    def forward(self, x, *args, _native_out: Optional[Tensor] = None, **kwargs):
        if _native_out is None:
            return _standalone(self, x)                        # attn(x) called directly (models/global_local_cl.py:41,45)
        return _native_out


def Upsample(dim: int, dim_out: Optional[int] = None) -> nn.Sequential:
    """models/unet_model.py:39-44."""
    return nn.Sequential(nn.Upsample(scale_factor=2, mode='nearest'), nn.Conv2d(dim, dim_out or dim, 3, padding=1))


def Downsample(dim: int, dim_out: Optional[int] = None) -> nn.Conv2d:
    """models/unet_model.py:47-49."""
    return nn.Conv2d(dim, dim_out or dim, 4, 2, 1)


class _Native:
    """One native handle of a Unet facade and what the facade knows about its state."""
    __slots__ = ('handle', 'device', 'fingerprint', 'bwd_fingerprint', 'bound')

    def __init__(self):
        self.handle = None
        self.device = None
        self.fingerprint = None          # (data_ptr, _version, dtype) of every bound tensor + precision the packs were made from
        self.bwd_fingerprint = None      # fingerprint the data-gradient packs were derived from
        self.bound: List[Tensor] = []    # keeps converted parameter copies alive


class _UnetTrainFn(torch.autograd.Function):
    """autograd bridge: forward = tedm_unet_forward_train, backward = tedm_unet_backward.  Gradients are
    returned as copies of the `grad_flat` views so that autograd's accumulation semantics hold."""

    @staticmethod
    def forward(ctx, unet, x, timestep, names, *params):
        ctx.unet, ctx.names = unet, names
        ctx.mask = [p.requires_grad for p in params]
        eps = unet.native_forward_train(x, timestep)
        ctx.generation = unet.tape_generation()         # the handle keeps ONE tape: a later forward overwrites it
        unet._fire_tap_hooks_from_tape()                # forward hooks on ups[i][2] fire on this route too (datasetDM_model.py:50-53)
        return eps

    @staticmethod
    def backward(ctx, d_eps):
        views = ctx.unet.native_backward(d_eps, generation=ctx.generation)
        grads = [views[k].clone() if m else None for k, m in zip(ctx.names, ctx.mask)]
        return (None, None, None, None, *grads)


class Unet(nn.Module):
    def __init__(self, dim: int = 64, init_dim: Optional[int] = None, out_dim: Optional[int] = None,
                 dim_mults: Sequence[int] = (1, 2, 4, 8), channels: int = 1, resnet_block_groups: int = 8,
                 learned_variance: bool = False, learned_sinusoidal_cond: bool = False,
                 learned_sinusoidal_dim: int = 16, **kwargs):
        super().__init__()
        self.channels = channels
        init_dim = init_dim if init_dim is not None else dim
        dim_mults = tuple(int(m) for m in dim_mults)
        self.init_conv = nn.Conv2d(channels, init_dim, 7, padding=3)
        dims = [init_dim, *[dim * m for m in dim_mults]]
        in_out = list(zip(dims[:-1], dims[1:]))
        time_dim = dim * 4
        self.learned_sinusoidal_cond = bool(learned_sinusoidal_cond)             # models/unet_model.py:278-292
        if self.learned_sinusoidal_cond:
            pos_emb, fourier_dim = LearnedSinusoidalPosEmb(learned_sinusoidal_dim), learned_sinusoidal_dim + 1
        else:
            pos_emb, fourier_dim = SinusoidalPosEmb(dim), dim
        self.time_mlp = nn.Sequential(pos_emb, nn.Linear(fourier_dim, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))

        def rb(a, b):
            return ResnetBlock(a, b, time_emb_dim=time_dim, groups=resnet_block_groups)

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        n = len(in_out)
        for ind, (d_in, d_out) in enumerate(in_out):
            last = ind >= n - 1
            self.downs.append(nn.ModuleList([
                rb(d_in, d_in), rb(d_in, d_in), Residual(PreNorm(d_in, LinearAttention(d_in))),
                Downsample(d_in, d_out) if not last else nn.Conv2d(d_in, d_out, 3, padding=1)]))
        mid = dims[-1]
        self.mid_block1 = rb(mid, mid)
        self.mid_attn = Residual(PreNorm(mid, Attention(mid)))
        self.mid_block2 = rb(mid, mid)
        for ind, (d_in, d_out) in enumerate(reversed(in_out)):
            last = ind == n - 1
            self.ups.append(nn.ModuleList([
                rb(d_out + d_in, d_out), rb(d_out + d_in, d_out), Residual(PreNorm(d_out, LinearAttention(d_out))),
                Upsample(d_out, d_in) if not last else nn.Conv2d(d_out, d_in, 3, padding=1)]))
        self.out_dim = out_dim if out_dim is not None else channels * (1 if not learned_variance else 2)
        self.final_res_block = rb(dim * 2, dim)
        self.final_conv = nn.Conv2d(dim, self.out_dim, 1)

        # ---- native side -----------------------------------------------------------------
        self._cfg = dict(dim=dim, init_dim=init_dim, out_dim=self.out_dim, dim_mults=dim_mults, channels=channels,
                         groups=resnet_block_groups, heads=4, dim_head=32,
                         learned_sinusoidal_dim=int(learned_sinusoidal_dim) if learned_sinusoidal_cond else 0)
        # Two native handles, one per ROLE: 'infer' (forward / module_forward, packed for `precision`) and 'train'
        # (forward_train / backward, packed for `train_precision`, owns the tape and the data-gradient packs).  Each owns its
        # packed weights, so an eval / sampling / validation forward between a training forward and its backward can neither
        # re-encode the packs the outstanding tape needs nor clear the backward packs.
        self._native: Dict[str, _Native] = {'infer': _Native(), 'train': _Native()}
        self._workspace: Optional[Tensor] = None
        self._train_ws: Optional[Tensor] = None
        # Inference batches of >= `stream_min_batch` images can be cut into `streams` sub-batches that run on that many side
        # HIP streams (one workspace each, the same handle and packed weights): the tails and launch gaps of one sub-batch's kernels
        # are filled by the other's.  Results are bit-identical (the UNet has no cross-sample op).  Measured on MI355X at
        # 32x256x256: 2 streams -3.8 % (16.19 -> 15.58 ms), 3 streams the same, 4 streams +5 % (sub-batches too small).
        self.streams = int(os.environ.get('TEDM_STREAMS') or kwargs.get('streams') or 1)
        self.stream_min_batch = 16
        self._side: List[Tuple] = []            # [(torch.cuda.Stream, workspace tensor or None)] per device, lazily
        self.grad_flat: Optional[Tensor] = None     # one flat fp32 buffer holding every parameter gradient
        self._grad_views: Dict[str, Tensor] = {}
        self._grad_handle = None
        # Arithmetic mode of the MFMA contractions (tedm_hip.h TEDM_PREC_*).  Module default 'bf16x3' for inference AND training
        # (eps 2e-5 from the fp32 reference; gradients within 1e-4 per tensor of fp32 autograd).  The fast inference mode
        # 'f16f4' (fp16 MFMA + one block-scaled fp4 MFMA for the cross terms, a single fp16 product inside the ResnetBlocks below
        # full resolution: eps 2-4.5e-4, gate 1e-3, 1.7x faster) is an explicit opt-in -- `Unet(..., precision='f16f4')`,
        # `config.precision`, `model.precision = 'f16f4'` or TEDM_PREC -- which bench.py takes and gates on its own eps check
        # (its margin is measured on synthetic weights only).  Inference and training use separate native handles,
        # so differing modes do not repack each other's weights.
        self.precision = os.environ.get('TEDM_PREC') or kwargs.get('precision') or 'bf16x3'
        self.train_precision = os.environ.get('TEDM_TRAIN_PREC') or kwargs.get('train_precision') or os.environ.get('TEDM_PREC') or 'bf16x3'
        self._module_ws: Optional[Tensor] = None
        for name, mod in self.named_modules():                  # back-reference for stand-alone calls (not a registered sub-module)
            if isinstance(mod, (ResnetBlock, Residual)):
                object.__setattr__(mod, '_owner', (self,))
                object.__setattr__(mod, '_qualname', name)
        self.last_trace: Dict[str, Tensor] = {}

    def invalidate_packed(self) -> None:
        """Force the packed (MFMA-fragment) weight copies to be rebuilt at the next forward.

        Repacking is keyed on (data_ptr, _version, dtype) of every parameter, which catches optimizer steps,
        `load_state_dict` and `.to()`.  In-place edits that bypass autograd's version counter -- `p.data.copy_()`,
        `p.data.clamp_()`, EMA `p.data.lerp_()`, `set_()`, external kernels writing through the pointer -- do NOT bump
        `_version`: call this after them, or the forward mixes stale packed conv weights with the live biases / norm
        parameters it reads through the bound pointers."""
        for n in self._native.values():
            n.fingerprint = None
            n.bwd_fingerprint = None

    # -- native handle ---------------------------------------------------------------------
    def __del__(self):
        try:
            for n in getattr(self, '_native', {}).values():
                if n.handle:
                    _lib.lib().tedm_unet_destroy(n.handle)
                    n.handle = None
        except Exception:
            pass

    @property
    def _handle(self):
        """The inference-role handle (trace / module queries)."""
        return self._native['infer'].handle

    def _ensure_handle(self, device: torch.device, role: str = 'infer') -> '_Native':
        L = _lib.lib()
        n = self._native[role]
        if n.handle is not None and n.device == device:
            return n
        if n.handle is not None:
            L.tedm_unet_destroy(n.handle)
            n.handle = None
        c = self._cfg
        cfg = _lib.UnetCfg()
        cfg.dim, cfg.init_dim, cfg.out_dim = c['dim'], c['init_dim'], c['out_dim']
        cfg.n_mults = len(c['dim_mults'])
        for i, m in enumerate(c['dim_mults']):
            cfg.dim_mults[i] = m
        cfg.channels, cfg.groups, cfg.heads, cfg.dim_head = c['channels'], c['groups'], c['heads'], c['dim_head']
        cfg.learned_sinusoidal_dim = c['learned_sinusoidal_dim']
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(L.tedm_unet_create(C.byref(cfg), C.byref(h)), 'tedm_unet_create')
        n.handle, n.device, n.fingerprint, n.bwd_fingerprint, n.bound = h, device, None, None, []
        return n

    def _native_tensors(self) -> Dict[str, Tensor]:
        sd = {k: v for k, v in self.named_parameters()}
        if not self.learned_sinusoidal_cond:
            sd['time_mlp.0.freqs'] = self.time_mlp[0].freqs
        return sd

    def _prec_name(self, training: bool) -> str:
        name = self.train_precision if training else self.precision
        if name not in _lib.PREC_BY_NAME:
            raise ValueError(f'unknown precision {name!r} (use one of {sorted(_lib.PREC_BY_NAME)})')
        return name

    def _sync_params(self, device: torch.device, stream_ptr: int, training: bool = False) -> '_Native':
        """(Re)bind parameter pointers of the role's handle and repack when anything changed (load_state_dict,
        optimizer step, .to(), arithmetic mode).  Returns the role's native state."""
        L = _lib.lib()
        nat = self._ensure_handle(device, 'train' if training else 'infer')
        tensors = self._native_tensors()
        prec = self._prec_name(training)
        fp = tuple((t.data_ptr(), t._version, t.dtype) for t in tensors.values()) + (prec,)
        if fp == nat.fingerprint:
            return nat
        nat.bound = []
        nat.bwd_fingerprint = None              # the pack below re-encodes what the data-gradient packs were derived from
        handle = nat.handle
        n = L.tedm_unet_num_params(handle)
        want = {L.tedm_unet_param_key(handle, i).decode() for i in range(n)}
        if want != set(tensors):
            raise _lib.TedmError(f'parameter set mismatch between facade and native plan: '
                                 f'{sorted(want ^ set(tensors))[:8]}')
        for k, t in tensors.items():
            if t.device != device:
                raise RuntimeError(f'parameter {k} is on {t.device}, input on {device}')
            tt = t.detach()
            if tt.dtype != torch.float32 or not tt.is_contiguous():
                tt = tt.float().contiguous()
            nat.bound.append(tt)
            _lib.check(L.tedm_unet_set_param(handle, k.encode(), tt.data_ptr(), tt.numel()), f'set_param({k})')
        _lib.check(L.tedm_unet_set_precision(handle, _lib.PREC_BY_NAME[prec]), 'set_precision')
        _lib.check(L.tedm_unet_pack(handle, stream_ptr), 'tedm_unet_pack')
        nat.fingerprint = fp
        return nat

    def _flags(self, taps_only: bool, trace: bool, training: bool = False) -> int:
        f = _lib.PREC_BY_NAME[self._prec_name(training)] << _lib.FWD_PREC_SHIFT
        if taps_only:
            f |= _lib.FWD_TAPS_ONLY
        if trace:
            f |= _lib.FWD_TRACE
        return f

    def _tap_shapes(self, B: int, H: int, W: int):
        c = self._cfg
        dims = [c['init_dim'], *[c['dim'] * m for m in c['dim_mults']]]
        n = len(c['dim_mults'])
        out = []
        for i in range(n):
            lvl = n - 1 - i                       # ups[i] runs at encoder level n-1-i
            f = 2 ** min(lvl, n - 1)
            out.append((B, H // f, W // f, dims[n - i]))
        return out

    # -- the hot path ----------------------------------------------------------------------
    def native_forward(self, x: Tensor, timestep: Optional[Tensor], *, want_taps: bool = False,
                       taps_only: bool = False, trace: bool = False):
        """Run the HIP forward. Returns (eps or None, [4 taps as NCHW-shaped channels_last views] or None)."""
        if not x.is_cuda:
            raise RuntimeError('tedm_amd.Unet runs on a ROCm GPU only: there is no CPU fallback '
                               '(use oracle/ for a CPU restatement in tests)')
        if x.dim() != 4 or x.shape[1] != self.channels:
            raise ValueError(f'expected input (B,{self.channels},H,W), got {tuple(x.shape)}')
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError('tedm_amd.Unet: gradients w.r.t. the input image are not computed '
                                      '(the reference never asks for them); detach x')
        L = _lib.lib()
        dev = x.device
        B, _, H, W = x.shape
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            handle = self._sync_params(dev, stream).handle
            xin = x.detach().float().contiguous()
            tin = None
            if timestep is not None:
                tin = timestep.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
                if tin.numel() != B:
                    raise ValueError(f'timestep must have {B} elements, got {tin.numel()}')
            flags = self._flags(taps_only, trace)
            eps = None if taps_only else torch.empty(B, self.out_dim, H, W, dtype=torch.float32, device=dev)
            taps = None
            if want_taps or taps_only:
                taps = [torch.empty(s, dtype=torch.float32, device=dev) for s in self._tap_shapes(B, H, W)]
            nsplit = min(self.streams, B // max(1, self.stream_min_batch // 2)) if (self.streams > 1 and B >= self.stream_min_batch and not trace) else 1

            def launch(b0: int, b1: int, ws_holder, st_ptr: int):
                """enqueue images [b0, b1) on stream st_ptr with the workspace kept in ws_holder[0]"""
                nb = b1 - b0
                need = L.tedm_unet_workspace_bytes(handle, nb, H, W, flags)
                if need == 0:
                    _lib.check(-1, f'workspace sizing for input {(nb,) + tuple(x.shape[1:])}')
                ws = ws_holder[0]
                if ws is None or ws.numel() < need or ws.device != dev:
                    ws_holder[0] = None
                    ws = ws_holder[0] = torch.empty(need, dtype=torch.uint8, device=dev)
                fp = None
                if taps is not None:
                    fp = (C.c_void_p * len(taps))(*[t[b0:b1].data_ptr() for t in taps])
                _lib.check(L.tedm_unet_forward(
                    handle, xin[b0:b1].data_ptr(), tin[b0:b1].data_ptr() if tin is not None else None, nb, H, W,
                    eps[b0:b1].data_ptr() if eps is not None else None, fp, ws.data_ptr(), ws.numel(), st_ptr, flags), 'tedm_unet_forward')

            if nsplit <= 1:
                holder = [self._workspace]
                launch(0, B, holder, stream)
                self._workspace = holder[0]
            else:
                # sub-batches on side streams: fork from / join into the caller's stream, so every tensor involved is ordered
                # on the caller's stream before the fork and after the join (no record_stream bookkeeping needed)
                cur = torch.cuda.current_stream(dev)
                while len(self._side) < nsplit:
                    self._side.append((torch.cuda.Stream(dev), [None]))
                if self._side[0][0].device != dev:
                    self._side = [(torch.cuda.Stream(dev), [None]) for _ in range(nsplit)]
                cuts = [B * i // nsplit for i in range(nsplit + 1)]
                for i in range(nsplit):
                    side, holder = self._side[i]
                    side.wait_stream(cur)
                    launch(cuts[i], cuts[i + 1], holder, side.cuda_stream)
                for i in range(nsplit):
                    cur.wait_stream(self._side[i][0])
            if trace:
                self.last_trace = self._read_trace()
            if taps is not None:
                taps = [t.permute(0, 3, 1, 2) for t in taps]     # NCHW-shaped views of the NHWC buffers
        return eps, taps

    # -- stand-alone sub-module calls ----------------------------------------------------------
    @torch.no_grad()
    def module_forward(self, name: str, x: Tensor, time_emb: Optional[Tensor] = None) -> Tensor:
        """Forward of ONE ResnetBlock (`block(x, t)`, t = time embedding or None) or Residual(PreNorm(attention)) block
        (`attn(x)`) through `tedm_unet_module_forward`.  NCHW in, NCHW-shaped channels-last view out.  Inference only."""
        if not x.is_cuda:
            raise RuntimeError('tedm_amd.Unet runs on a ROCm GPU only: there is no CPU fallback')
        L = _lib.lib()
        dev = x.device
        B, Cx, H, W = x.shape
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            self._sync_params(dev, stream)
            cin, cout = C.c_int(), C.c_int()
            _lib.check(L.tedm_unet_module_channels(self._handle, name.encode(), C.byref(cin), C.byref(cout)), f'module {name}')
            if Cx != cin.value:
                raise ValueError(f'{name} takes {cin.value} input channels, got {Cx}')
            xin = x.detach().float().permute(0, 2, 3, 1).contiguous()               # NHWC
            st = None
            if time_emb is not None:
                st = torch.nn.functional.silu(time_emb.detach().float()).contiguous()   # ResnetBlock.mlp's SiLU (:149-152)
            need = L.tedm_unet_module_workspace_bytes(self._handle, name.encode(), B, H, W)
            if need == 0:
                _lib.check(-1, f'module workspace sizing for {name} at {tuple(x.shape)}')
            if self._module_ws is None or self._module_ws.numel() < need or self._module_ws.device != dev:
                self._module_ws = None
                self._module_ws = torch.empty(need, dtype=torch.uint8, device=dev)
            out = torch.empty(B, H, W, cout.value, dtype=torch.float32, device=dev)
            _lib.check(L.tedm_unet_module_forward(self._handle, name.encode(), xin.data_ptr(), cin.value, None, 0,
                                                  st.data_ptr() if st is not None else None, B, H, W, out.data_ptr(),
                                                  self._module_ws.data_ptr(), self._module_ws.numel(), stream,
                                                  self._flags(False, False)), f'tedm_unet_module_forward({name})')
        return out.permute(0, 3, 1, 2)

    # -- training path ------------------------------------------------------------------------
    def grad_views(self, device: torch.device) -> Dict[str, Tensor]:
        """Per-parameter views into `grad_flat` (allocated and bound to the native handle on first use)."""
        L = _lib.lib()
        handle = self._ensure_handle(device, 'train').handle
        params = dict(self.named_parameters())
        if self.grad_flat is None or self.grad_flat.device != device:
            total = sum(p.numel() for p in params.values())
            self.grad_flat = torch.zeros(total, dtype=torch.float32, device=device)
            self._grad_views, off = {}, 0
            for k, n in self.grad_layout():          # reverse backward-completion order: the buffer finishes from its end (dp.py)
                self._grad_views[k] = self.grad_flat[off:off + n].view(params[k].shape)
                off += n
            self._grad_handle = None
        if self._grad_handle is not handle:
            for k, v in self._grad_views.items():
                _lib.check(L.tedm_unet_set_grad(handle, k.encode(), v.data_ptr(), v.numel()), f'set_grad({k})')
            self._grad_handle = handle
        return self._grad_views

    def grad_layout(self):
        """[(key, numel)] in the order the parameters' gradients are laid out in `grad_flat` (dp.grad_layout)."""
        from .dp import grad_layout
        return grad_layout([(k, p.numel()) for k, p in self.named_parameters()], len(self._cfg['dim_mults']))

    def native_forward_train(self, x: Tensor, timestep: Optional[Tensor]) -> Tensor:
        """HIP forward that records the tape for `native_backward` in the training workspace."""
        if not x.is_cuda:
            raise RuntimeError('tedm_amd.Unet runs on a ROCm GPU only: there is no CPU fallback')
        if x.dim() != 4 or x.shape[1] != self.channels:
            raise ValueError(f'expected input (B,{self.channels},H,W), got {tuple(x.shape)}')
        L = _lib.lib()
        dev = x.device
        B, _, H, W = x.shape
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            nat = self._sync_params(dev, stream, training=True)
            handle = nat.handle
            if nat.bwd_fingerprint != nat.fingerprint:
                _lib.check(L.tedm_unet_pack_backward(handle, stream), 'tedm_unet_pack_backward')
                nat.bwd_fingerprint = nat.fingerprint
            self.grad_views(dev)
            self._train_x = x.detach().float().contiguous()          # the stem's weight gradient reads it
            tin = None
            if timestep is not None:
                tin = timestep.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
                if tin.numel() != B:
                    raise ValueError(f'timestep must have {B} elements, got {tin.numel()}')
            need = L.tedm_unet_train_workspace_bytes(handle, B, H, W)
            if need == 0:
                _lib.check(-1, f'training workspace sizing for input {tuple(x.shape)}')
            if self._train_ws is None or self._train_ws.numel() < need or self._train_ws.device != dev:
                self._train_ws = None
                self._train_ws = torch.empty(need, dtype=torch.uint8, device=dev)
            eps = torch.empty(B, self.out_dim, H, W, dtype=torch.float32, device=dev)
            _lib.check(L.tedm_unet_forward_train(
                handle, self._train_x.data_ptr(), tin.data_ptr() if tin is not None else None, B, H, W,
                eps.data_ptr(), self._train_ws.data_ptr(), self._train_ws.numel(), stream, self._flags(False, False, training=True)),
                'tedm_unet_forward_train')
        return eps

    def _taps_hooked(self) -> bool:
        return any(len(lvl[2]._forward_hooks) or len(lvl[2]._forward_pre_hooks) for lvl in self.ups)

    def _fire_tap_hooks_from_tape(self) -> None:
        """Training / autograd route: the hooked tensors are what the tape keeps as the tail convs' inputs.  The hooks receive
        DETACHED channels-last views into the training workspace (valid until the backward of this forward or the next training
        forward; a hook that keeps one must copy it, as the reference's own `save_activations` does with .detach().cpu())."""
        if not self._taps_hooked():
            return
        L = _lib.lib()
        h = self._native['train'].handle
        for i, lvl in enumerate(self.ups):
            off, shp = C.c_size_t(), (C.c_int * 4)()
            _lib.check(L.tedm_unet_tape_tap(h, i, C.byref(off), shp), 'tedm_unet_tape_tap')
            b, hh, ww, c = list(shp)
            tap = self._train_ws[off.value: off.value + 4 * b * hh * ww * c].view(torch.float32).view(b, hh, ww, c).permute(0, 3, 1, 2)
            lvl[2](tap, _native_out=tap)

    def tape_generation(self) -> int:
        """Generation of the training tape the native handle holds (bumped by every `native_forward_train`)."""
        h = self._native['train'].handle
        return int(_lib.lib().tedm_unet_tape_generation(h)) if h is not None else 0

    def native_backward(self, d_eps: Tensor, generation: int = 0, on_group=None) -> Dict[str, Tensor]:
        """Backward of the last `native_forward_train`; overwrites `grad_flat` and returns its views.  `generation`
        (from `tape_generation()` right after the forward) makes the call fail loudly -- 'tape overwritten by a later
        forward' -- when another training forward ran in between (the handle keeps one tape).  `on_group(name)` is invoked
        each time every gradient of one top-level module has been enqueued (order: final_conv, final_res_block, ups.3..0, mid_block2,
        mid_attn, mid_block1, downs.3..0, init_conv, time_mlp)."""
        L = _lib.lib()
        dev = d_eps.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            g = d_eps.detach().float().contiguous()
            handle = self._native['train'].handle
            cb, errs = None, []
            if on_group is not None:                     # called from inside the C backward each time a top-level module's gradients
                def _mark(_user, name):                  # are enqueued (tedm_hip.h: tedm_unet_set_backward_callback)
                    try:
                        on_group(name.decode())
                    except BaseException as e:           # never let an exception cross the C frame
                        errs.append(e)
                cb = _lib.BACKWARD_MARK_FN(_mark)
                _lib.check(L.tedm_unet_set_backward_callback(handle, cb, None), 'set_backward_callback')
            try:
                _lib.check(L.tedm_unet_backward_gen(handle, g.data_ptr(), self._train_ws.data_ptr(), self._train_ws.numel(),
                                                    stream, int(generation)), 'tedm_unet_backward')
            finally:
                if cb is not None:
                    L.tedm_unet_set_backward_callback(handle, None, None)
            if errs:
                raise errs[0]
        return self._grad_views

    @torch.no_grad()
    def fp16_range_report(self, x: Tensor, timestep: Optional[Tensor] = None, limit: float = 65504.0) -> Dict[str, float]:
        """max |activation| of every traced module output on this input, computed with the fp32-range arithmetic (bf16x3).

        The fast inference mode `f16f4` carries the convolution operands as fp16 (+ fp4 corrections): an activation beyond
        +-65504 SATURATES there and everything downstream is wrong without any error (measured: a model whose residual stream
        reaches 3e5 gives eps 0.87 off under f16f4 and 6e-5 under bf16x3, profiles/r04_eps_margin_trained.txt).  Networks with
        a sane initialisation stay orders of magnitude below the limit; call this once per checkpoint on a representative batch.
        Raises if `precision == 'f16f4'` and any module exceeds `limit` / 2."""
        keep = self.precision
        try:
            self.precision = 'bf16x3'
            self.native_forward(x, timestep, trace=True)
            rep = {k: float(v.abs().max()) for k, v in self.last_trace.items()}
        finally:
            self.precision = keep
        worst = max(rep.values()) if rep else 0.0
        if keep == 'f16f4' and worst > limit / 2:
            k = max(rep, key=rep.get)
            raise OverflowError(f"precision 'f16f4' cannot carry this model: |activation| reaches {worst:.3g} at {k} (fp16 operands saturate at "
                                f"{limit:g}); use precision='bf16x3'")
        return rep

    def _read_trace(self) -> Dict[str, Tensor]:
        L = _lib.lib()
        out = {}
        for i in range(L.tedm_unet_trace_count(self._handle)):
            name = L.tedm_unet_trace_name(self._handle, i).decode()
            off = C.c_size_t()
            shp = (C.c_int * 4)()
            _lib.check(L.tedm_unet_trace_info(self._handle, i, C.byref(off), shp), 'trace_info')
            b, h, w, c = list(shp)
            n = b * h * w * c
            flat = self._workspace[off.value: off.value + 4 * n].view(torch.float32)
            out[name] = flat.view(b, h, w, c).permute(0, 3, 1, 2).clone()
        return out

    def forward(self, x: Tensor, timestep: Optional[Tensor] = None, cond: Optional[Tensor] = None) -> Tensor:
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if x.requires_grad:
                raise NotImplementedError('tedm_amd.Unet: gradients w.r.t. the input image are not computed')
            names, params = zip(*self.named_parameters())
            return _UnetTrainFn.apply(self, x, timestep, names, *params)
        hooked = self._taps_hooked()
        eps, taps = self.native_forward(x, timestep, want_taps=hooked)
        if hooked:
            for lvl, tap in zip(self.ups, taps):
                lvl[2](tap, _native_out=tap)          # fires the registered forward hooks
        return eps
