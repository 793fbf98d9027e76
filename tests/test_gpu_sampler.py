"""GPU (MI355X): the sampling-step kernel (tedm_amd/csrc/sampler.hip; reference models/diffusion_model.py:205-267) against the
torch ops it replaces -- the threshold against torch.quantile on the same |x0| (same two order statistics, value within 1e-6),
x0 / posterior mean / x_{t-1} against the reference formulas."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev(built_lib):
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def run_step(dm, x_t, out, t, noise, objective, clip=True):
    """direct call through the C ABI, thresholds included"""
    from tedm_amd import _lib
    L = _lib.lib()
    B, chw = x_t.shape[0], x_t[0].numel()
    ranks = np.float32(0.995) * np.float32(chw - 1)
    k_lo = int(np.floor(ranks)); w = float(np.float32(ranks - np.float32(k_lo)))
    x0, mean, prev = torch.empty_like(x_t), torch.empty_like(x_t), torch.empty_like(x_t)
    s = torch.empty(B, device=x_t.device)
    p = lambda v: None if v is None else v.data_ptr()
    _lib.check(L.tedm_sampler_step(x_t.data_ptr(), out.data_ptr(), t.data_ptr(), p(noise), dm.sqrt_recip_alphas_cumprod.data_ptr(),
                                   dm.sqrt_recipm1_alphas_cumprod.data_ptr(), dm.posterior_mean_coef1.data_ptr(),
                                   dm.posterior_mean_coef2.data_ptr(), dm.posterior_log_variance_clipped.data_ptr(), 1000, B, chw,
                                   objective, 1 if clip else 0, k_lo, w, x0.data_ptr(), mean.data_ptr(), prev.data_ptr(), s.data_ptr(),
                                   torch.cuda.current_stream(x_t.device).cuda_stream), 'tedm_sampler_step')
    torch.cuda.synchronize()
    return x0, mean, prev, s


# (5,1,256,256) and the small shapes take the register-resident selection (<= 64 K values per sample), (3,2,256,192) = 96 K values the
# LDS-histogram kernel kept for larger samples
@pytest.mark.parametrize('shape', [(5, 1, 256, 256), (3, 3, 40, 24), (2, 1, 8, 8), (3, 2, 256, 192)])
def test_sampler_step_kernel_vs_torch_ops(dev, shape):
    from argparse import Namespace
    from tedm_amd import DiffusionModel
    from tedm_amd.diffusion import _take
    dm = DiffusionModel(Namespace(normalize=True, dim=32)).to(dev)
    g = torch.Generator().manual_seed(3)
    B = shape[0]
    x_t = torch.randn(*shape, generator=g)
    out = torch.randn(*shape, generator=g)
    scale = torch.tensor([3.0, 0.2, 1.0, 7.0, 0.9][:B]).view(B, 1, 1, 1)          # thresholds above and below the clamp at 1
    x_t, out = x_t * scale, out * scale
    if B > 2:                                                                     # heavy ties: the two order statistics coincide
        out[2] = torch.round(out[2] * 2) / 2
        x_t[2] = torch.round(x_t[2] * 2) / 2
    t = torch.tensor([999, 500, 0, 17, 250][:B])
    noise = torch.randn(*shape, generator=g)
    x_t, out, t, noise = x_t.to(dev), out.to(dev), t.to(dev), noise.to(dev)
    for objective in (0, 1):
        x0_raw = (_take(dm.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - _take(dm.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * out) if objective == 0 else out
        s_ref = torch.quantile(x0_raw.flatten(1).abs(), 0.995, dim=1)
        s_c = torch.clamp(s_ref, min=1.0)[:, None, None, None]
        x0_ref = torch.clip(x0_raw, -s_c, s_c) / s_c
        mean_ref = _take(dm.posterior_mean_coef1, t, x_t.shape) * x0_ref + _take(dm.posterior_mean_coef2, t, x_t.shape) * x_t
        prev_ref = mean_ref + (0.5 * _take(dm.posterior_log_variance_clipped, t, x_t.shape)).exp() * noise * (t > 0).float().view(-1, 1, 1, 1)
        x0, mean, prev, s = run_step(dm, x_t, out, t, noise, objective)
        assert float((s - s_c.flatten()).abs().max() / s_c.abs().max()) < 1e-6, (s, s_c.flatten())
        for a, b in ((x0, x0_ref), (mean, mean_ref), (prev, prev_ref)):
            assert float((a - b).abs().max()) < 2e-6 * max(1.0, float(b.abs().max()))
    # clip_denoised = False: no threshold
    x0, mean, _, s = run_step(dm, x_t, out, t, None, 0, clip=False)
    x0_raw = _take(dm.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - _take(dm.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * out
    assert float((x0 - x0_raw).abs().max()) < 2e-6 * float(x0_raw.abs().max()) and bool((s == 1).all())
