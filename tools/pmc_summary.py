#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv (+ kernel_trace.csv for durations) per kernel name.
usage: python tools/pmc_summary.py <dir-with-csvs> [name-filter]"""
import csv, glob, re, sys
from collections import defaultdict
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set); tns = defaultdict(float)
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*', '', n); return n.replace(' ', '')
for r in csv.DictReader(open(cc)):
    n = short(r['Kernel_Name'])
    if flt and flt not in n: continue
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in cnt[n]:
        cnt[n].add(r['Dispatch_Id']); tns[n] += dur.get(r['Dispatch_Id'], 0)
for n in sorted(agg, key=lambda k: -tns[k])[:14]:
    c = agg[n]; k = len(cnt[n])
    print(f'{n}  launches={k} avg_us={tns[n]/k/1e3:.1f}')
    print('   ' + '  '.join(f'{a}={v/k:.4g}' for a, v in sorted(c.items())))
    if 'SQ_WAVE_CYCLES' in c:
        wc = c['SQ_WAVE_CYCLES']
        print('   frac of wave-cycles: wait_any=%.2f wait_inst_any=%.2f active=%.2f ; mfma_busy/busy_cycles=%.3f' % (
            c['SQ_WAIT_ANY']/wc, c['SQ_WAIT_INST_ANY']/wc, c['SQ_ACTIVE_INST_ANY']/wc,
            c['SQ_VALU_MFMA_BUSY_CYCLES']/max(c['SQ_BUSY_CYCLES'],1)))
    if 'GRBM_GUI_ACTIVE' in c:
        print('   eff clock GHz = %.3f' % (c['GRBM_GUI_ACTIVE']/8/tns[n]))
