#!/usr/bin/env python3
"""Experiment: does running the batch as sub-batches on several HIP streams (kernel tails / launch gaps of one sub-batch filled by
the others) beat one batch on one stream?  N Unet facades with identical weights = N handles / workspaces.
usage: python tools/two_stream_probe.py [prec]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch
from tedm_amd import Unet, load_synth_weights

dev = torch.device('cuda:0')
B, S = 32, 256
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16f4'
ms = []
for _ in range(4):
    m = Unet(64, precision=prec)
    load_synth_weights(m, seed=0)
    ms.append(m.to(dev).eval())
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 1, S, S, generator=g).to(dev)
t = torch.randint(0, 1000, (B,), generator=g).to(dev)
streams = [torch.cuda.Stream(dev) for _ in range(4)]


def one():
    return ms[0](x, t)


def multi(cuts):
    cur = torch.cuda.current_stream(dev)
    outs, b0 = [], 0
    for i, n in enumerate(cuts):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            outs.append(ms[i](x[b0:b0 + n], t[b0:b0 + n]))
        b0 += n
    for i in range(len(cuts)):
        cur.wait_stream(streams[i])
    return outs


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = one()
    print(prec, 'bit-identical:', torch.equal(torch.cat(multi([16, 16])), ref))
    print(f'one stream, B=32: {timeit(one):.2f} ms')
    for cuts in ([16, 16], [11, 11, 10], [8, 8, 8, 8], [20, 12], [32, 32]):
        if sum(cuts) == 64:
            fn = lambda: [one_on(i) for i in range(2)]
            def two_full():
                cur = torch.cuda.current_stream(dev)
                for i in range(2):
                    streams[i].wait_stream(cur)
                    with torch.cuda.stream(streams[i]):
                        ms[i](x, t)
                for i in range(2):
                    cur.wait_stream(streams[i])
            print(f'two streams, each the FULL batch (two steps overlapped): {timeit(two_full) / 2:.2f} ms per step')
        else:
            print(f'{len(cuts)} streams {cuts}: {timeit(lambda: multi(cuts)):.2f} ms')
