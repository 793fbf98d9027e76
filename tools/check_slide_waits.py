#!/usr/bin/env python3
"""Static check of the sliding-window convolution's band loop (tedm_amd/csrc/conv_slide.hip) in the compiled ISA.

The band loop consumes, in iteration b + 1, registers that buffer loads of iteration b filled.  The compiler's own waitcnt pass
does not cover that (it emits either no vmcnt wait in the loop or `vmcnt(0)` at the first inline asm), so the kernel waits
explicitly with hand-counted `s_waitcnt vmcnt(N)`.  This script proves the count from the disassembly, for every kernel instance:
for each `buffer_load` destination register, walking forward from the load around the loop's back edge to the first instruction
that reads it, there must be an `s_waitcnt vmcnt(N)` with N <= (vector-memory instructions issued since the load, up to the wait).
It also reports the loop's instruction mix (what the MFMA gaps have to hide).

    python tools/check_slide_waits.py            # compiles conv_slide.hip to assembly with hipcc (no GPU needed)
"""
import collections
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
VMEM = ('buffer_load', 'buffer_store', 'global_load', 'global_store', 'global_atomic', 'buffer_atomic')


def regs(tok):
    """VGPR indices named by one operand token: v12, v[4:7], |v3|, -v3"""
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def split_ops(line):
    op, _, rest = line.partition(' ')
    toks = [t.strip() for t in rest.split(',')] if rest else []
    return op, toks


def band_loops(asm, kernel):
    """[(kernel name, [instructions of a band loop])]: every basic block with >= 100 MFMAs (conv3x3_slide128_kernel has two,
    one per wave role)"""
    out = []
    for m in re.finditer(r'^(_ZN\S*' + kernel + r'\S*):.*?\n(.*?)\.end_amdhsa_kernel', asm, re.S | re.M):
        blocks, cur = [], []
        for l in m.group(2).split('\n'):
            s = l.strip()
            if re.match(r'^\.LBB\d+_\d+:', s):
                blocks.append(cur)
                cur = []
            elif s and not s.startswith(';') and not s.startswith('.'):
                cur.append(s.split(';')[0].strip())
        blocks.append(cur)
        for b in blocks:
            if sum('v_mfma' in x for x in b) >= 100:
                out.append((m.group(1), b))
    return out


def check(name, loop):
    n = len(loop)
    bad = []
    loads = [i for i, l in enumerate(loop) if l.startswith('buffer_load')]
    for i in loads:
        op, toks = split_ops(loop[i])
        dst = set(regs(toks[0]))
        vm_since, worst = 0, None
        ok = False
        for step in range(1, n + 1):                       # forward, around the back edge, at most one full turn
            l = loop[(i + step) % n]
            o, t = split_ops(l)
            if o == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', l)
                if m and int(m.group(1)) <= vm_since:
                    ok = True
                    break
                continue
            if o.startswith(VMEM):
                srcs = t if ('store' in o) else t[1:]
                if dst & set(r for x in srcs for r in regs(x)):
                    break
                vm_since += 1
                continue
            srcs = t[1:] if (o.startswith('v_') or o.startswith('ds_read')) and not o.startswith('v_cmp') else t
            if o.startswith('ds_write') or o.startswith('v_cmp') or o.startswith('v_mfma'):
                srcs = t if not o.startswith('v_mfma') else t[1:]
            if dst & set(r for x in srcs for r in regs(x)):
                break
        if not ok:
            bad.append((i, loop[i], vm_since))
    mix = collections.Counter()
    for l in loop:
        o = l.split()[0]
        mix['mfma' if 'mfma' in o else 'valu' if o.startswith('v_') else 'lds' if o.startswith('ds_') else 'vmem' if o.startswith(VMEM)
            else 'waitcnt' if o == 's_waitcnt' else 'nop' if o == 's_nop' else 'salu'] += 1
    return loads, bad, mix


def main():
    rc = 0
    # (source, kernel, expected (MFMAs, loop-carried loads) per band loop)
    for fname, kernel, expect in (('conv_slide.hip', 'conv3x3_slide64_kernel', {(216, 18)}), ('conv_slide128.hip', 'conv3x3_slide128_kernel', {(108, 12), (108, 6)})):
        src = ROOT / 'tedm_amd' / 'csrc' / fname
        with tempfile.TemporaryDirectory() as td:
            out = Path(td) / 'k.s'
            cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-gpu-rdc', '--cuda-device-only', '-S',
                   str(src), '-o', str(out)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                print(r.stderr)
                return 2
            asm = out.read_text()
        loops = band_loops(asm, kernel)
        if not loops:
            print(f'no {kernel} band loop found')
            return 2
        for name, loop in loops:
            loads, bad, mix = check(name, loop)
            nm = mix['mfma']
            print(f'{name}: band loop {len(loop)} instructions, {nm} MFMA, per MFMA gap: ' +
                  ', '.join(f'{k} {v / max(nm, 1):.2f}' for k, v in sorted(mix.items()) if k != 'mfma') + f'; {len(loads)} loop-carried loads')
            if (nm, len(loads)) not in expect:
                print(f'  UNEXPECTED band structure: (MFMAs, loads) = {(nm, len(loads))}, expected one of {sorted(expect)}')
                rc = 1
            for i, l, vm in bad:
                print(f'  NOT COVERED: {l} (instruction {i}): first use reached with {vm} vector-memory instructions in between and no vmcnt wait <= that')
                rc = 1
    print('ok' if rc == 0 else 'FAILED')
    return rc


if __name__ == '__main__':
    sys.exit(main())
