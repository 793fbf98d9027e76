"""ORACLE -- test infrastructure only, never shipped on the product path.

CPU restatement (plain torch fp32 functional ops, NCHW) of the TEDM diffusion-UNet
hot path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this module; the product (`tedm_amd`) must never route through it.

Parity status: PINNED -- validated in the build container against the imported
reference (`oracle/make_golden.py`, reference at /root/reference) and against the
golden vectors committed under `tests/golden/` (tests/test_oracle_golden.py).

Every function cites the reference file:line it restates.  Weights arrive as a flat
dict keyed like the reference `Unet.state_dict()`.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# schedules / q-sample                         reference: models/diffusion_model.py
# --------------------------------------------------------------------------------------
def cosine_betas(timesteps: int, s: float = 0.008) -> Tensor:
    """models/diffusion_model.py:32-47 -- cosine schedule, fp32, clipped to [0, 0.999]."""
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float32)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def linear_betas(timesteps: int, start: float = 1e-4, end: float = 2e-2) -> Tensor:
    """models/diffusion_model.py:16-29."""
    scale = 1000 / timesteps
    return torch.linspace(scale * start, scale * end, timesteps, dtype=torch.float32)


def diffusion_buffers(timesteps: int = 1000, beta_schedule: str = 'cosine',
                      p2_gamma: float = 0.0, p2_k: float = 1.0) -> Dict[str, Tensor]:
    """models/diffusion_model.py:75-115 -- the nine per-timestep coefficient buffers."""
    if beta_schedule == 'cosine':
        betas = cosine_betas(timesteps)
    elif beta_schedule == 'linear':
        betas = linear_betas(timesteps)
    else:
        raise ValueError(f'unknown beta schedule {beta_schedule}')
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        'sqrt_alphas_cumprod': torch.sqrt(ac),
        'sqrt_recip_alphas_cumprod': torch.sqrt(1.0 / ac),
        'sqrt_recipm1_alphas_cumprod': torch.sqrt(1.0 / ac - 1),
        'sqrt_one_minus_alphas_cumprod': torch.sqrt(1.0 - ac),
        'posterior_variance': post_var,
        'posterior_log_variance_clipped': torch.log(post_var.clamp(min=1e-20)),
        'posterior_mean_coef1': betas * torch.sqrt(ac_prev) / (1.0 - ac),
        'posterior_mean_coef2': (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        'p2_loss_weight': (p2_k + ac / (1 - ac)) ** (-p2_gamma),
    }


def q_sample(bufs: Dict[str, Tensor], x0: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """models/diffusion_model.py:176-203 + trainers/utils.py:48-59:
    x_t = sqrt(acp[t]) * x0 + sqrt(1-acp[t]) * noise, coefficients gathered per sample."""
    a = bufs['sqrt_alphas_cumprod'].gather(-1, t).reshape(-1, 1, 1, 1)
    b = bufs['sqrt_one_minus_alphas_cumprod'].gather(-1, t).reshape(-1, 1, 1, 1)
    return a * x0 + b * noise


# --------------------------------------------------------------------------------------
# UNet pieces                                   reference: models/unet_model.py
# --------------------------------------------------------------------------------------
def sinusoidal_emb(t: Tensor, dim: int) -> Tensor:
    """models/unet_model.py:76-93 -- [sin(t f_i), cos(t f_i)], f_i = exp(-i ln(1e4)/(half-1))."""
    half = dim // 2
    f = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    e = t[:, None] * f[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def learned_sinusoidal_emb(t: Tensor, weights: Tensor) -> Tensor:
    """models/unet_model.py:96-114 (LearnedSinusoidalPosEmb.forward): cat(t, sin(f), cos(f)), f = t * weights * 2 * pi."""
    t = t[:, None]
    freqs = t * weights[None, :] * 2 * math.pi
    return torch.cat((t, freqs.sin(), freqs.cos()), dim=-1)


def time_mlp(p: Params, t: Tensor, dim: int) -> Tensor:
    """models/unet_model.py:278-292 -- (learned) sinusoid -> Linear -> exact-erf GELU -> Linear."""
    e = learned_sinusoidal_emb(t, p['time_mlp.0.weights']) if 'time_mlp.0.weights' in p else sinusoidal_emb(t, dim)
    e = F.linear(e, p['time_mlp.1.weight'], p['time_mlp.1.bias'])
    e = F.gelu(e)
    return F.linear(e, p['time_mlp.3.weight'], p['time_mlp.3.bias'])


def chan_layernorm(x: Tensor, g: Tensor) -> Tensor:
    """models/unet_model.py:52-61 -- per-pixel LayerNorm over channels, gain only,
    biased variance, eps = 1e-5 (the reference evaluates the fp32 branch)."""
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def block(p: Params, pre: str, x: Tensor, groups: int,
          scale_shift: Optional[Tuple[Tensor, Tensor]]) -> Tensor:
    """models/unet_model.py:119-135 -- conv3x3 -> GroupNorm -> FiLM (x*(scale+1)+shift) -> SiLU."""
    x = F.conv2d(x, p[pre + '.proj.weight'], p[pre + '.proj.bias'], padding=1)
    x = F.group_norm(x, groups, p[pre + '.norm.weight'], p[pre + '.norm.bias'], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def resnet_block(p: Params, pre: str, x: Tensor, temb: Optional[Tensor], groups: int) -> Tensor:
    """models/unet_model.py:138-175."""
    ss = None
    if temb is not None and (pre + '.time_mlp.1.weight') in p:
        e = F.linear(F.silu(temb), p[pre + '.time_mlp.1.weight'], p[pre + '.time_mlp.1.bias'])
        e = e[:, :, None, None]
        ss = e.chunk(2, dim=1)
    h = block(p, pre + '.block1', x, groups, ss)
    h = block(p, pre + '.block2', h, groups, None)
    if (pre + '.res_conv.weight') in p:
        x = F.conv2d(x, p[pre + '.res_conv.weight'], p[pre + '.res_conv.bias'])
    return h + x


def linear_attention(p: Params, pre: str, x: Tensor, heads: int = 4, dim_head: int = 32) -> Tensor:
    """models/unet_model.py:178-210 wrapped by Residual(PreNorm(.)) :29-36,:64-73.
    `pre` is the Residual module prefix (e.g. 'downs.0.2')."""
    b, c, h, w = x.shape
    n = h * w
    y = chan_layernorm(x, p[pre + '.fn.norm.g'])
    qkv = F.conv2d(y, p[pre + '.fn.fn.to_qkv.weight'])
    q, k, v = (t.reshape(b, heads, dim_head, n) for t in qkv.chunk(3, dim=1))
    q = q.softmax(dim=-2) * dim_head ** -0.5
    k = k.softmax(dim=-1)
    v = v / n
    ctx = torch.einsum('bhdn,bhen->bhde', k, v)
    out = torch.einsum('bhde,bhdn->bhen', ctx, q).reshape(b, heads * dim_head, h, w)
    out = F.conv2d(out, p[pre + '.fn.fn.to_out.0.weight'], p[pre + '.fn.fn.to_out.0.bias'])
    out = chan_layernorm(out, p[pre + '.fn.fn.to_out.1.g'])
    return out + x


def full_attention(p: Params, pre: str, x: Tensor, heads: int = 4, dim_head: int = 32,
                   scale: float = 16.0) -> Tensor:
    """models/unet_model.py:213-241 wrapped by Residual(PreNorm(.)).  NB q and k are
    l2-normalised along the TOKEN axis (F.normalize(dim=-1) on (b,h,d,n), :23,:234)."""
    b, c, h, w = x.shape
    n = h * w
    y = chan_layernorm(x, p[pre + '.fn.norm.g'])
    qkv = F.conv2d(y, p[pre + '.fn.fn.to_qkv.weight'])
    q, k, v = (t.reshape(b, heads, dim_head, n) for t in qkv.chunk(3, dim=1))
    q = F.normalize(q, dim=-1)
    k = F.normalize(k, dim=-1)
    sim = torch.einsum('bhdi,bhdj->bhij', q, k) * scale
    attn = sim.softmax(dim=-1)
    out = torch.einsum('bhij,bhdj->bhid', attn, v)            # (b, h, n, d)
    out = out.permute(0, 1, 3, 2).reshape(b, heads * dim_head, h, w)
    out = F.conv2d(out, p[pre + '.fn.fn.to_out.weight'], p[pre + '.fn.fn.to_out.bias'])
    return out + x


def unet_forward(p: Params, x: Tensor, t: Optional[Tensor], *, dim: int = 64,
                 dim_mults: Sequence[int] = (1, 2, 4, 8), groups: int = 8,
                 return_taps: bool = False, return_trace: bool = False):
    """models/unet_model.py:333-368.  Returns eps (B,out_dim,H,W); with return_taps also the
    four hooked decoder-attention outputs (models/datasetDM_model.py:50-53); with
    return_trace a dict of named intermediates for per-layer parity tests."""
    nres = len(dim_mults)
    trace: Dict[str, Tensor] = {}
    x = F.conv2d(x, p['init_conv.weight'], p['init_conv.bias'], padding=3)
    r = x
    trace['init_conv'] = x
    temb = time_mlp(p, t.float(), dim) if t is not None else None
    if temb is not None:
        trace['temb'] = temb
    hs: List[Tensor] = []
    for i in range(nres):
        pre = f'downs.{i}'
        x = resnet_block(p, pre + '.0', x, temb, groups)
        trace[pre + '.0'] = x
        hs.append(x)
        x = resnet_block(p, pre + '.1', x, temb, groups)
        trace[pre + '.1'] = x
        x = linear_attention(p, pre + '.2', x)
        trace[pre + '.2'] = x
        hs.append(x)
        wgt = p[pre + '.3.weight']
        if wgt.shape[-1] == 4:      # Downsample: Conv 4x4 stride 2 pad 1 (:47-49)
            x = F.conv2d(x, wgt, p[pre + '.3.bias'], stride=2, padding=1)
        else:                       # last level: Conv 3x3 pad 1 (:307-308)
            x = F.conv2d(x, wgt, p[pre + '.3.bias'], padding=1)
        trace[pre + '.3'] = x
    x = resnet_block(p, 'mid_block1', x, temb, groups)
    trace['mid_block1'] = x
    x = full_attention(p, 'mid_attn', x)
    trace['mid_attn'] = x
    x = resnet_block(p, 'mid_block2', x, temb, groups)
    trace['mid_block2'] = x
    taps: List[Tensor] = []
    for i in range(nres):
        pre = f'ups.{i}'
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(p, pre + '.0', x, temb, groups)
        trace[pre + '.0'] = x
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(p, pre + '.1', x, temb, groups)
        trace[pre + '.1'] = x
        x = linear_attention(p, pre + '.2', x)
        trace[pre + '.2'] = x
        taps.append(x)
        if (pre + '.3.1.weight') in p:   # Upsample: nearest x2 then Conv3x3 (:39-44)
            x = F.interpolate(x, scale_factor=2, mode='nearest')
            x = F.conv2d(x, p[pre + '.3.1.weight'], p[pre + '.3.1.bias'], padding=1)
        else:                            # last level: Conv3x3 (:323-324)
            x = F.conv2d(x, p[pre + '.3.weight'], p[pre + '.3.bias'], padding=1)
        trace[pre + '.3'] = x
    x = torch.cat((x, r), dim=1)
    x = resnet_block(p, 'final_res_block', x, temb, groups)
    trace['final_res_block'] = x
    eps = F.conv2d(x, p['final_conv.weight'], p['final_conv.bias'])
    out = (eps,)
    if return_taps:
        out = out + (taps,)
    if return_trace:
        out = out + (trace,)
    return out[0] if len(out) == 1 else out


# --------------------------------------------------------------------------------------
# wrappers                     reference: models/diffusion_model.py, models/datasetDM_model.py
# --------------------------------------------------------------------------------------
def diffusion_forward(p: Params, bufs: Dict[str, Tensor], x0: Tensor, t: Tensor, noise: Tensor,
                      normalize: bool = True, **kw) -> Tensor:
    """models/diffusion_model.py:158-174 with the noise made explicit: optional x*2-1,
    q-sample, UNet."""
    if normalize:
        x0 = x0 * 2 - 1
    return unet_forward(p, q_sample(bufs, x0, t, noise), t, **kw)


def train_loss(p: Params, bufs: Dict[str, Tensor], x0: Tensor, t: Tensor, noise: Tensor,
               normalize: bool = True, **kw) -> Tensor:
    """models/diffusion_model.py:120-143 (objective pred_noise): L1 -> per-sample mean ->
    x p2_loss_weight[t] -> batch mean."""
    eps = diffusion_forward(p, bufs, x0, t, noise, normalize, **kw)
    loss = (eps - noise).abs().flatten(1).mean(dim=1)
    return (loss * bufs['p2_loss_weight'].gather(-1, t)).mean()


def p_mean_variance(p: Params, bufs: Dict[str, Tensor], x_t: Tensor, t: Tensor, percentile: float = 0.995,
                    clip_denoised: bool = True, **kw) -> Tuple[Tensor, Tensor, Tensor]:
    """models/diffusion_model.py:221-267 (objective pred_noise): eps -> x0 (:269-285), per-sample
    `percentile` dynamic thresholding (:224-231), posterior mean / clipped log-variance (:259-267)."""
    eps = unet_forward(p, x_t, t, **kw)
    take = lambda name: bufs[name].gather(-1, t).reshape(-1, 1, 1, 1)
    x0 = take('sqrt_recip_alphas_cumprod') * x_t - take('sqrt_recipm1_alphas_cumprod') * eps
    if clip_denoised:
        s = torch.quantile(x0.flatten(1).abs(), percentile, dim=1)
        s = torch.max(s, torch.tensor(1.0))[:, None, None, None]
        x0 = torch.clip(x0, -s, s) / s
    mean = take('posterior_mean_coef1') * x0 + take('posterior_mean_coef2') * x_t
    return mean, take('posterior_log_variance_clipped'), x0


def extract_features(p: Params, bufs: Dict[str, Tensor], x0: Tensor, steps: Sequence[int],
                     noise: Tensor, **kw) -> Tensor:
    """models/datasetDM_model.py:67-83 -- for every t in steps: q-sample the UN-normalised
    x0 (:76), run the UNet, nearest-upsample the 4 decoder taps to (S,S) (:80-81), concat
    over channels in order [t0:(tap0..tap3), t1:...]."""
    acts = []
    S = x0.shape[-1]
    for ts in steps:
        t = torch.full((x0.shape[0],), int(ts), dtype=torch.long)
        _, taps = unet_forward(p, q_sample(bufs, x0, t, noise), t, return_taps=True, **kw)
        for f in taps:
            acts.append(F.interpolate(f, size=[S, S]))
    return torch.cat(acts, dim=1)


def pixel_classifier(pc: Params, feats: Tensor, training: bool = False) -> Tensor:
    """models/datasetDM_model.py:57-64 -- Conv1x1 -> ReLU -> BN -> Conv1x1 -> ReLU -> BN -> Conv1x1.
    `pc` keys: '0.weight','0.bias','2.weight','2.bias','2.running_mean','2.running_var',
    '3.*','5.*','6.*' (nn.Sequential indices)."""
    x = F.relu(F.conv2d(feats, pc['0.weight'], pc['0.bias']))
    x = F.batch_norm(x, pc['2.running_mean'], pc['2.running_var'], pc['2.weight'], pc['2.bias'],
                     training=training, eps=1e-5)
    x = F.relu(F.conv2d(x, pc['3.weight'], pc['3.bias']))
    x = F.batch_norm(x, pc['5.running_mean'], pc['5.running_var'], pc['5.weight'], pc['5.bias'],
                     training=training, eps=1e-5)
    return F.conv2d(x, pc['6.weight'], pc['6.bias'])


# --------------------------------------------------------------------------------------
# TEDM: shared-weights classifier, ensembling, metrics
#       reference: trainers/train_datasetDM.py, trainers/train_baseline.py, app.py,
#                  auxiliary/postprocessing/testing_shared_weights.py
# --------------------------------------------------------------------------------------
def tedm_classifier(pc: Params, feats: Tensor, n_steps: int, training: bool = False) -> Tensor:
    """trainers/train_datasetDM.py:30-42 -- Rearrange('b (step act) h w -> (b step) act h w') followed by the SAME
    Conv1x1(960,128) -> ReLU -> BN -> Conv1x1(128,32) -> ReLU -> BN -> Conv1x1(32,1) for every timestep.
    `pc` keys are the nn.Sequential indices behind the Rearrange: '1.*', '3.*', '4.*', '6.*', '7.*'.  Output
    (B*n_steps, 1, S, S), row index b*n_steps + step.  With training=True BatchNorm uses the statistics of the
    (B*n_steps) pixel batch (single-device semantics of the reference)."""
    B, CT, H, W = feats.shape
    x = feats.reshape(B, n_steps, CT // n_steps, H, W).reshape(B * n_steps, CT // n_steps, H, W)
    shifted = {f'{int(k.split(".", 1)[0]) - 1}.{k.split(".", 1)[1]}': v for k, v in pc.items()}
    return pixel_classifier(shifted, x, training=training)


def tedm_ensemble(logits: Tensor, n_steps: int) -> Tuple[Tensor, Tensor]:
    """auxiliary/postprocessing/testing_shared_weights.py:113-133 (and app.py:79 for one image): sigmoid, regroup
    '(b step) 1 h w -> step b 1 h w', mean over the timesteps, threshold at 0.5.  Returns (probability, mask)."""
    p = torch.sigmoid(logits)
    p = p.reshape(-1, n_steps, *p.shape[1:]).transpose(0, 1).mean(0)
    return p, p > .5


def replicate_targets(y: Tensor, n_steps: int) -> Tensor:
    """trainers/train_baseline.py:30 -- repeat(y, 'b c h w -> (b step) c h w')."""
    return y.repeat_interleave(n_steps, dim=0)


def bce_loss(pred: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
    """trainers/train_baseline.py:44-45 -- BCE-with-logits, mean over (h, w) per (b, c), then mean.  Returns
    (loss, expanded_loss (b, c))."""
    e = F.binary_cross_entropy_with_logits(pred, y, reduction='none').mean(dim=(2, 3))
    return e.mean(), e


def dice(x_hat: Tensor, x: Tensor) -> Tensor:
    """trainers/train_baseline.py:146-149, per (b, c); 0/0 -> nan as in the reference."""
    inter = torch.logical_and(x_hat, x).sum(dim=(2, 3))
    return 2 * inter / (x_hat.sum(dim=(2, 3)) + x.sum(dim=(2, 3)))


def precision(x_hat: Tensor, x: Tensor) -> Tensor:
    """trainers/train_baseline.py:151-155."""
    tp = torch.logical_and(x, x_hat).sum(dim=(2, 3))
    fp = torch.logical_and(1 - x, x_hat).sum(dim=(2, 3))
    return tp / (tp + fp)


def recall(x_hat: Tensor, x: Tensor) -> Tensor:
    """trainers/train_baseline.py:157-161."""
    tp = torch.logical_and(x, x_hat).sum(dim=(2, 3))
    fn = torch.logical_and(x, ~x_hat).sum(dim=(2, 3))
    return tp / (tp + fn)
