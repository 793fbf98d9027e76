#!/usr/bin/env python3
"""Per-layer-shape timing of the convolution launches of one UNet forward (32x1x256x256): HIP-event profiler at level 3
(one entry per kernel symbol + Cin->Cout @HxW).  usage: python tools/conv_shape_profile.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from argparse import Namespace
from tedm_amd import DiffusionModel, _lib, load_synth_weights
L=_lib.lib()
dev=torch.device('cuda:0')
dm=DiffusionModel(Namespace(normalize=True, dim=64, precision=os.environ.get('TEDM_PREC', 'f16f4'))); load_synth_weights(dm.model, seed=0); dm=dm.to(dev).eval()  # the mode bench.py times (the module default is bf16x3)
print(f'precision {dm.model.precision}')
g=torch.Generator().manual_seed(0)
x=torch.rand(32,1,256,256,generator=g).to(dev); t=torch.randint(0,1000,(32,),generator=g).to(dev)
with torch.no_grad():
    for _ in range(3): dm.model(x,t)
    torch.cuda.synchronize()
    L.tedm_profile_reset(); L.tedm_profile_enable(3)
    for _ in range(5): dm.model(x,t)
    torch.cuda.synchronize(); L.tedm_profile_collect(); L.tedm_profile_enable(0)
ents=_lib.profile_entries()
ents=[e for e in ents if 'conv' in e[0]]
ents.sort(key=lambda e:-e[2])
tot=sum(e[2] for e in ents)
for name,n,ms,fl,by in ents:
    print(f'{ms/tot*100:5.1f}% {ms/n*1e3:8.1f} us x{n//5:3d}  {fl/(ms*1e-3)/1e12:6.1f} TF/s  {name}')
