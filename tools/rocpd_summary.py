#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`):
time per step, share, launches per step, average duration.  usage: rocpd_summary.py DB STEPS_IN_RUN [TOP]"""
import re
import sqlite3
import sys


def short(n: str) -> str:
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\((?:[^()]|\([^()]*\))*\)\s*(?:\[clone[^\]]*\])?$', '', n)   # argument list
    return n


def main():
    db, steps = sys.argv[1], float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    c = sqlite3.connect(db)
    rows = c.execute('select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc').fetchall()
    tot = sum(r[2] for r in rows)
    print(f'total kernel time {tot:.2f} ms over {steps:g} steps = {tot / steps:.2f} ms/step')
    print(f'{"ms/step":>9} {"share":>6} {"n/step":>7} {"avg_us":>9}  kernel')
    for n, k, ms, avg in rows[:top]:
        print(f'{ms / steps:9.3f} {100 * ms / tot:5.1f}% {k / steps:7.1f} {avg:9.1f}  {short(n)[:150]}')


if __name__ == '__main__':
    main()
