#!/usr/bin/env python3
"""Phase timeline of conv_igemm_kernel from the diagnostic build (tools/build_diag.sh):
   TEDM_HIP_LIB=tedm_amd/libtedm_diag.so python tools/conv_trace.py [Cin Cout S B]
Runs one 3x3 conv through tedm_op_conv2d with TEDM_TRACE_FILE set and prints, per phase, the median cycles over
waves: prologue, MFMA segment of each chunk, LDS staging, barrier wait, epilogue; plus how the workgroups that shared
a CU overlapped."""
import os
import sys
import ctypes as C

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))


def main():
    cin, cout, S, B = (int(v) for v in (sys.argv[1:5] + ['64', '64', '256', '32'][len(sys.argv) - 1:]))
    tf = '/tmp/conv_trace.bin'
    os.environ['TEDM_TRACE_FILE'] = tf
    from tedm_amd import _lib
    L = _lib.lib()
    if os.environ.get('CONV_SOLO'):
        _lib.check(L.tedm_set_tuning(b'conv_solo', int(os.environ['CONV_SOLO'])))
    if os.environ.get('CONV_PRIO'):
        _lib.check(L.tedm_set_tuning(b'conv_prio', int(os.environ['CONV_PRIO'])))
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, S, S, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.zeros(cout, device=dev)
    out = torch.empty(B, S, S, cout, device=dev)
    ne = C.c_int(0)
    _lib.check(L.tedm_op_conv2d(x.data_ptr(), cin, None, 0, B, S, S, w.data_ptr(), b.data_ptr(), cout, 3, 1, 0, None, None, None, None,
                                None, None, None, out.data_ptr(), None, 0, C.byref(ne), 0, None), 'conv')
    t = np.fromfile(tf, dtype=np.uint64).reshape(2048, 4, 64).astype(np.int64)
    ok = t[:, :, 0] > 0
    print(f'conv {cin}->{cout} {S}x{S} B={B}: traced waves {int(ok.sum())}')
    nc = cin // 16
    rows = []
    for blk in range(2048):
        for wv in range(4):
            e = t[blk, wv]
            if e[0] == 0 or e[3] == 0:
                continue
            d = {'pro': e[2] - e[0], 'total': e[3] - e[0]}
            for it in range(min(nc, 14)):
                d[f'mfma{it}'] = e[5 + 4 * it] - e[4 + 4 * it]
                d[f'stage{it}'] = e[6 + 4 * it] - e[5 + 4 * it]
                d[f'bar{it}'] = e[7 + 4 * it] - e[6 + 4 * it]
            d["epi"] = e[3] - e[7 + 4 * (min(nc, 14) - 1)]
            d['hw'] = int(e[1])
            d['start'] = e[0]
            rows.append(d)
    keys = ['pro'] + [f'{p}{i}' for i in range(min(nc, 14)) for p in ('mfma', 'stage', 'bar')] + ['epi', 'total']
    for k in keys:
        v = np.array([r[k] for r in rows])
        print(f'  {k:8s} median {np.median(v):9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles')
    tot = np.median([r['total'] for r in rows])
    mf = sum(np.median([r[f'mfma{i}'] for r in rows]) for i in range(min(nc, 14)))
    print(f'  ideal MFMA per wave alone: {nc * 108 * 32} cycles ({nc} chunks x 108 MFMAs x 32); measured MFMA segments {mf:.0f}; total {tot:.0f}')


if __name__ == '__main__':
    main()
