#!/usr/bin/env python3
"""Tuning harness (GPU): time the convolution kernel on the UNet's representative layer shapes at the benchmark batch (32 images),
per arithmetic mode and tile.  usage: python tools/conv_bench.py [--prec 0,3] [--tiles default|all] [--shapes a,b,...]"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.getcwd())
from tedm_amd import _lib

SHAPES = {   # name: (k, stride, ups, C1, C2, Cout, S, gn)
    'l0_64_64':      (3, 1, 0, 64, 0, 64, 256, 0),
    'l0_64_64_gn':   (3, 1, 0, 64, 0, 64, 256, 1),
    'l0_128_64':     (3, 1, 0, 64, 64, 64, 256, 0),
    'l1_192_128':    (3, 1, 0, 128, 64, 128, 128, 0),
    'l1_128_128_gn': (3, 1, 0, 128, 0, 128, 128, 1),
    'l2_256_256_gn': (3, 1, 0, 256, 0, 256, 64, 1),
    'l3_512_512_gn': (3, 1, 0, 512, 0, 512, 32, 1),
    'l3_768_512':    (3, 1, 0, 512, 256, 512, 32, 0),
    'l2_384_256':    (3, 1, 0, 256, 128, 256, 64, 0),
    'up_512_256':    (3, 1, 1, 512, 0, 256, 32, 0),
    'up3_512_256':   (3, 1, 3, 512, 0, 256, 32, 0),
    'up3_256_128':   (3, 1, 3, 256, 0, 128, 64, 0),
    'dn_256_256':    (4, 2, 0, 256, 0, 256, 64, 0),
    'l3_256_512':    (3, 1, 0, 256, 0, 512, 32, 0),
    'dn_64_64':      (4, 2, 0, 64, 0, 64, 256, 0),
    'res_128_64':    (1, 1, 0, 64, 64, 64, 256, 0),
}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--prec', default='0,3')
    ap.add_argument('--tiles', default='default')
    ap.add_argument('--shapes', default=','.join(SHAPES))
    ap.add_argument('--B', type=int, default=32)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--tune', default='', help='key=value,... passed to tedm_set_tuning')
    a = ap.parse_args()
    L = _lib.lib()
    dev = torch.device('cuda:0')
    for kv in filter(None, a.tune.split(',')):
        k_, v_ = kv.split('=')
        _lib.check(L.tedm_set_tuning(k_.encode(), int(v_)), 'tune')
    g = torch.Generator().manual_seed(0)
    p = lambda t: None if t is None else t.data_ptr()
    for name in a.shapes.split(','):
        k, stride, ups, C1, C2, Cout, S, gn = SHAPES[name]
        B = a.B
        x1 = torch.randn(B, S, S, C1, generator=g).to(dev)
        x2 = torch.randn(B, S, S, C2, generator=g).to(dev) if C2 else None
        w = (torch.randn(Cout, C1 + C2, k, k, generator=g) / ((C1 + C2) * k * k) ** 0.5).to(dev)
        b = torch.randn(Cout, generator=g).to(dev)
        So = S // 2 if stride == 2 else (2 * S if ups else S)
        out = torch.empty(B, So, So, Cout, device=dev)
        ga = gb = None
        if gn:
            ga = (1 + 0.1 * torch.randn(B, C1 + C2, generator=g)).to(dev); gb = (0.1 * torch.randn(B, C1 + C2, generator=g)).to(dev)
        E = L.tedm_op_conv2d_stats_entries(Cout, k, stride, So, So)
        st = torch.zeros(B, E, Cout, 2, device=dev) if ups != 3 else None
        flops = 2.0 * B * So * So * Cout * (C1 + C2) * k * k
        byts = 4.0 * B * (S * S * (C1 + C2) + So * So * Cout)
        for prec in map(int, a.prec.split(',')):
            tiles = [None] if a.tiles == 'default' else ([0, 2, 3] if Cout >= 128 else [1, 3])
            for tile in tiles:
                if tile is not None:
                    _lib.check(L.tedm_set_tuning(b'conv_tile_wide' if Cout >= 128 else b'conv_tile_narrow', tile), 'tile')
                ms = C.c_float()
                rc = L.tedm_op_conv2d_bench(p(x1), C1, p(x2), C2, B, S, S, p(w), p(b), Cout, k, stride, ups, p(ga), p(gb), None,
                                            p(out), p(st), 1, prec, a.iters, C.byref(ms), torch.cuda.current_stream(dev).cuda_stream)
                if rc != 0:
                    print(f'{name:16s} prec {prec} tile {tile}: rc {rc} {L.tedm_last_error().decode()}')
                    continue
                print(f'{name:16s} prec {prec} tile {str(tile):4s} {ms.value * 1e3:8.1f} us  {flops / ms.value / 1e9:7.1f} TF/s  '
                      f'{byts / ms.value / 1e9:6.2f} TB/s(alg)', flush=True)
        if a.tiles != 'default':
            L.tedm_set_tuning(b'conv_tile_wide', 2); L.tedm_set_tuning(b'conv_tile_narrow', 1)

if __name__ == '__main__':
    main()
