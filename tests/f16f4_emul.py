"""CPU emulation (test infrastructure) of the TEDM_PREC_F16F4 operand encoding of tedm_amd/csrc/common.h:
fp16 hi + MX-fp4 (e2m1, E8M0 scale per 32 channels) residual and copy; products Ah*Bh + Al*B4 + A4*Bl."""
import torch
import torch.nn.functional as F


def _fp4(x, scale):
    """e2m1 round-to-nearest-even of x / scale, saturating at 6, times scale"""
    y = x / scale
    ay = y.abs().clamp(min=1e-30)
    e = torch.floor(torch.log2(ay)).clamp(min=0.0)          # 1 - bias = 0: subnormal step 0.5
    step = torch.exp2(e - 1)
    return (torch.round(y / step) * step).clamp(-6.0, 6.0) * scale


def encode(x, dim):
    """x -> (hi, lo4, v4): blocks of 32 along `dim` (which must be a multiple of 32)"""
    xm = x.movedim(dim, -1)
    shp = xm.shape
    xb = xm.reshape(*shp[:-1], shp[-1] // 32, 32)
    amax = xb.abs().amax(dim=-1, keepdim=True)
    E = torch.floor(torch.log2(amax.clamp(min=2.0 ** -126))) + 127          # biased exponent
    E = torch.where(amax > 0, E, torch.zeros_like(E))
    e_v4 = (E - 2).clamp(min=1)
    e_lo = (torch.clamp(E - 11, min=102) - 2).clamp(min=1)
    hi = xb.clamp(-65504.0, 65504.0).half().float()
    lo4 = _fp4(xb - hi, torch.exp2(e_lo - 127))
    v4 = _fp4(xb, torch.exp2(e_v4 - 127))
    back = lambda t: t.reshape(shp).movedim(-1, dim)
    return back(hi), back(lo4), back(v4)


def conv2d(x, w, b=None, stride=1, padding=0, split=None):
    """x NCHW (zero padding is applied AFTER encoding, as the kernel stages it); split: channels of the first concat source"""
    xh, xl, x4 = encode(x.double(), 1)
    wh, wl, w4 = encode(w.double(), 1)
    c = lambda a, k: F.conv2d(a, k, None, stride, padding)
    y = c(xh, wh) + c(xl, w4) + c(x4, wl)
    return y if b is None else y + b.double().reshape(1, -1, 1, 1)
