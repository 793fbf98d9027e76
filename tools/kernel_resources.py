#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2>&1 | python tools/kernel_resources.py"""
import re, subprocess, sys
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r'remark: Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    for key, pat in (('vgpr', r'remark:\s+VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('spill', r'VGPRs Spill: (\d+)'),
                     ('sgpr', r'remark:\s+SGPRs: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'),
                     ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
    if 'error' in line: print(line.rstrip())
names = [r['name'] for r in rows]
try:
    dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
for r, d in zip(rows, dem):
    d = re.sub(r'\(anonymous namespace\)::', '', d)
    d = re.sub(r'^void ', '', d); d = re.sub(r'\(.*$', '', d); d = d.replace(' ', '')
    print(f"{d:58s} vgpr={r.get('vgpr')} agpr={r.get('agpr')} spill={r.get('spill')} scratch={r.get('scratch')} sgpr={r.get('sgpr')} occ={r.get('occ')} lds={r.get('lds')}")
