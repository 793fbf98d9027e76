#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of
`bench.py` into per-kernel HBM-side traffic per launch, corrected as MI355X_MICROARCH.md prescribes:
both counters are in KiB... (rocprofv3 reports kilobytes); on gfx950 FETCH_SIZE reports exactly 1/2 of
the bytes of wide coalesced reads (verified on this code base: ln_res reads 2 tensors and writes 1,
raw FETCH_SIZE == WRITE_SIZE), so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
usage: python tools/pmc_traffic.py <fetch-dir> <write-dir> <out.json>"""
import csv, glob, json, re, sys
from collections import defaultdict

def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*', '', n)
    return n.replace(' ', '')

def load(d, ctr):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg, cnt = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == ctr:
            n = short(r['Kernel_Name']); agg[n] += float(r['Counter_Value']); cnt[n].add(r['Dispatch_Id'])
    return {n: (agg[n] / len(cnt[n]), len(cnt[n])) for n in agg}

fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {}
for n, (f, k) in fetch.items():
    w = write.get(n, (0.0, 0))[0]
    out[n] = {'launches': k, 'fetch_size_kb_raw': round(f, 1), 'write_size_kb': round(w, 1),
              'hbm_bytes_per_launch': round((2 * f + w) * 1024)}
json.dump({'command': 'bench.py --steps 2 --warmup 1 --no-cpu-baseline', 'correction': 'bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024',
           'kernels': out}, open(sys.argv[3], 'w'), indent=1)
print(f'wrote {sys.argv[3]} ({len(out)} kernels)')
