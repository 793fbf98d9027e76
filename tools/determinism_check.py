"""Run the native forward twice with TEDM_FWD_TRACE and report the first module whose output differs."""
import sys, torch
sys.path.insert(0, '.')
from tedm_amd import Unet, load_synth_weights, _lib
B, S = int(sys.argv[1]), int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split('='); _lib.check(_lib.lib().tedm_set_tuning(k.encode(), int(v)))
dev = torch.device('cuda:0')
m = Unet(64); load_synth_weights(m, 0); m = m.to(dev).eval()
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 1, S, S, generator=g).to(dev); t = torch.randint(0, 1000, (B,), generator=g).to(dev)
with torch.no_grad():
    e1, _ = m.native_forward(x, t, trace=True); t1 = dict(m.last_trace)
    e2, _ = m.native_forward(x, t, trace=True); t2 = dict(m.last_trace)
bad = [(k, float((t1[k] - t2[k]).abs().max())) for k in t1 if not torch.equal(t1[k], t2[k])]
print('eps equal:', torch.equal(e1, e2), ' differing modules:', bad[:6])
