#!/usr/bin/env python3
"""Per-dispatch timeline of ONE forward step from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/trace_one_step.py DIR [steps_in_run]
Prints the dispatches of the last step in launch order (kernel, duration us, gap to the previous kernel us)."""
import csv, glob, re, sys


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\((?:[^()]|\([^()]*\))*\)\s*(?:\[clone[^\]]*\])?$', '', n)
    return n[:110]


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ks = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
    # a step starts at q_sample_kernel
    starts = [i for i, k in enumerate(ks) if k[0].startswith('q_sample_kernel')]
    if len(starts) < 2:
        print('no step boundaries found'); return
    a, b = starts[-2], starts[-1]
    tot = 0
    prev_end = None
    for name, s, e in ks[a:b]:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f'{(e - s) / 1e3:9.1f} us  gap {gap:6.1f}  {name}')
        tot += e - s
        prev_end = e
    print(f'kernel time {tot / 1e6:.3f} ms, wall {(ks[b][1] - ks[a][1]) / 1e6:.3f} ms, {b - a} dispatches')


if __name__ == '__main__':
    main()
