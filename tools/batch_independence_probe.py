#!/usr/bin/env python3
"""Which module output first depends on the batch composition?  Runs the traced forward on a batch and on a sub-batch and
reports, per traced module in execution order, whether the sub-batch rows are bit-identical.  usage: python tools/batch_independence_probe.py [S] [B] [b]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from tedm_amd import Unet, load_synth_weights

S, B, b = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 64), (2, 6), (3, 3)))
for prec in ('bf16x3', 'f16f4'):
    m = Unet(64, precision=prec)
    load_synth_weights(m, seed=0)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 1, S, S, generator=g).cuda()
    t = torch.tensor([50, 150, 250] * B)[:B].cuda()
    with torch.no_grad():
        m.native_forward(x, t, trace=True)
        full = dict(m.last_trace)
        m.native_forward(x[:b].contiguous(), t[:b].contiguous(), trace=True)
        part = dict(m.last_trace)
    bad = [(k, float((full[k][:b] - part[k]).abs().max())) for k in full if not torch.equal(full[k][:b], part[k])]
    print(prec, f'S={S} B={B} vs {b}:', 'all bit-identical' if not bad else f'{len(bad)} of {len(full)} differ; first: {bad[:4]}')
