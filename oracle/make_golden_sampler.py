"""ORACLE TOOLING (build container only, needs /root/reference): golden vectors for the ancestral sampler's
deterministic part -- `DiffusionModel.p_mean_variance` / `sample_timestep(t=0)` of the imported
reference (models/diffusion_model.py:205-267) at t in {999, 500, 0}, covering both branches of the dynamic
threshold (s > 1 and s clamped to 1).  Also checks oracle/unet_ref.py:p_mean_variance against the reference.

    python oracle/make_golden_sampler.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
from make_golden import GOLD, cfg_ns, import_reference, rel   # noqa: E402


def main():
    from oracle import unet_ref as O
    from tedm_amd.weights import synth_state_dict
    _, RDiff, _ = import_reference()
    torch.set_grad_enabled(False)
    dm = RDiff(cfg_ns(64))
    w = synth_state_dict({k: tuple(v.shape) for k, v in dm.model.state_dict().items()}, seed=0)
    dm.model.load_state_dict(w)
    dm.eval()
    g = torch.Generator().manual_seed(77)
    x_t = torch.randn(3, 1, 32, 32, generator=g)
    x_t[2] *= 0.2                                    # t = 0 sample: |x0| quantile < 1 -> threshold clamps to 1
    t = torch.tensor([999, 500, 0])
    mean, logvar, x0 = dm.p_mean_variance(x_t, t)
    s = torch.quantile(dm.model_predictions(x_t, t)[1].flatten(1).abs(), dm.dynamic_threshold_percentile, dim=1)
    print('thresholds s =', s.tolist())
    assert float(s[0]) > 1.0 and float(s[2]) < 1.0
    last = dm.sample_timestep(x_t[2:3], 0, None)      # t = 0: no noise is added
    bufs = O.diffusion_buffers()
    mo, lo, xo = O.p_mean_variance(w, bufs, x_t, t)
    e = max(rel(mo, mean), rel(xo, x0))
    print(f'oracle p_mean_variance vs reference rel-L2 {e:.3e}')
    assert e < 1e-5 and torch.equal(lo.expand_as(logvar), logvar)
    np.savez(GOLD / 'sampler_d64_s32.npz', x_t=x_t.numpy(), t=t.numpy(), mean=mean.numpy(),
             logvar=logvar.reshape(-1).numpy(), x0=x0.numpy(), s=s.numpy(), x_prev_t0=last.numpy())


if __name__ == '__main__':
    main()
