"""Fused linear-attention sweeps: reproducibility + parity vs a torch fp64 restatement of unet_model.py:197-207."""
import sys, torch
sys.path.insert(0, '.')
from tedm_amd import _lib
L = _lib.lib()
B, hw, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for kv in sys.argv[4:]:
    k_, v_ = kv.split('='); _lib.check(L.tedm_set_tuning(k_.encode(), int(v_)))
dev = torch.device('cuda:0')
g_ = torch.Generator().manual_seed(3)
x = torch.randn(B, hw, C, generator=g_)
gain = 1 + 0.1 * torch.randn(C, generator=g_)
w = torch.randn(384, C, generator=g_) / C ** 0.5
mean = x.mean(-1, keepdim=True); var = x.var(-1, unbiased=False, keepdim=True); rstd = (var + 1e-5).rsqrt()
stats = torch.cat((rstd, mean * rstd), -1).reshape(-1, 2).contiguous()
xd, gd, sd, wd = x.to(dev), gain.to(dev), stats.to(dev), w.reshape(384, C, 1, 1).contiguous().to(dev)
outs = []
for rep in range(3):
    kmax = torch.empty(B, 128, device=dev); ctx = torch.empty(B, 4, 32, 32, device=dev); o = torch.empty(B, hw, 128, device=dev)
    _lib.check(L.tedm_op_linattn_fused(xd.data_ptr(), C, gd.data_ptr(), sd.data_ptr(), wd.data_ptr(), B, hw, kmax.data_ptr(),
                                       ctx.data_ptr(), o.data_ptr(), 0, torch.cuda.current_stream().cuda_stream))
    outs.append((kmax.cpu(), ctx.cpu(), o.cpu()))
for nm, i in (('kmax', 0), ('ctx', 1), ('o', 2)):
    print(nm, 'reproducible:', [torch.equal(outs[0][i], outs[r][i]) for r in (1, 2)],
          'maxdiff', [float((outs[0][i] - outs[r][i]).abs().max()) for r in (1, 2)])
xn = ((x - mean) * rstd * gain).double()
qkv = xn @ w.double().t()
q, k, v = qkv[..., :128], qkv[..., 128:256], qkv[..., 256:]
def heads(t): return t.reshape(B, hw, 4, 32).permute(0, 2, 3, 1)      # b h d n
q, k, v = heads(q), heads(k), heads(v)
kref = k.amax(-1).reshape(B, 128)
ctx_ref = torch.einsum('bhdn,bhen->bhde', k.softmax(-1), v / hw)
o_ref = torch.einsum('bhde,bhdn->bhen', ctx_ref, q.softmax(-2) * 32 ** -0.5).permute(0, 3, 1, 2).reshape(B, hw, 128)
rel = lambda a, b: float((a.double() - b).norm() / b.norm())
print('rel err: kmax %.2e ctx %.2e o %.2e' % (rel(outs[0][0], kref), rel(outs[0][1], ctx_ref), rel(outs[0][2], o_ref)))
# ---- error structure of o
err = (outs[0][2].double() - o_ref).abs()                      # (B, hw, 128)
big = err > 1e-3 * o_ref.abs().max()
print('bad elements:', int(big.sum()), 'of', big.numel())
if big.any():
    idx = big.nonzero()
    print('bad per image:', torch.bincount(idx[:, 0], minlength=B).tolist())
    print('bad per head :', torch.bincount(idx[:, 2] // 32, minlength=4).tolist())
    print('bad per 128-px tile (first 40 tiles):', torch.bincount(idx[:, 1] // 128, minlength=hw // 128)[:40].tolist())
    print('bad per pixel-in-tile%32:', torch.bincount(idx[:, 1] % 32, minlength=32).tolist())
    print('bad per e (channel%32):', torch.bincount(idx[:, 2] % 32, minlength=32).tolist())
    print('first bad:', idx[:5].tolist(), [float(outs[0][2][tuple(i)]) for i in idx[:5]], [float(o_ref[tuple(i)]) for i in idx[:5]])
