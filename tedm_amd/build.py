"""Build libtedm_hip.so (gfx950) in-tree with hipcc.  No GPU is needed: hipcc cross-compiles.

    python -m tedm_amd.build [--force]

The library is rebuilt when any source under tedm_amd/csrc or include/ is newer than it.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / 'csrc'
OBJ = CSRC / 'build'
LIB = PKG / 'libtedm_hip.so'
SOURCES = ['conv_igemm_p0.hip', 'conv_igemm_p1.hip', 'conv_igemm_p2.hip', 'conv_igemm_p3.hip', 'conv_igemm_p4.hip', 'conv_igemm.hip', 'conv_slide.hip', 'conv_slide128.hip', 'pack.hip', 'elementwise.hip', 'sampler.hip', 'attention.hip', 'linattn_fused.hip', 'wgrad.hip', 'wgrad_mfma.hip', 'backward.hip', 'head_train.hip', 'attention_bwd.hip', 'unet_host.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _deps() -> list:
    return sorted(CSRC.glob('*.hip')) + sorted(CSRC.glob('*.h')) + sorted(CSRC.glob('*.inc')) + sorted((ROOT / 'include').glob('*.h'))


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in _deps())


def _compile(src: str) -> Path:
    obj = OBJ / (src + '.o')
    deps = [CSRC / src] + sorted(CSRC.glob('*.h')) + sorted(CSRC.glob('*.inc')) + sorted((ROOT / 'include').glob('*.h'))
    if obj.exists() and all(p.stat().st_mtime <= obj.stat().st_mtime for p in deps):
        return obj
    cmd = [_hipcc(), *FLAGS, '-c', str(CSRC / src), '-o', str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    OBJ.mkdir(parents=True, exist_ok=True)
    if force:
        for o in OBJ.glob('*.o'):
            o.unlink()
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(f'built {LIB} ({LIB.stat().st_size} bytes)')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
