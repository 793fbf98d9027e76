"""eps determinism without trace (arena reuse active)."""
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
    outs = [m(x, t).clone() for _ in range(4)]
print(B, S, sys.argv[3:], [torch.equal(outs[0], o) for o in outs[1:]], [float((outs[0]-o).abs().max()) for o in outs[1:]])
