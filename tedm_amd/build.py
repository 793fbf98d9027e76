"""Build libtedm_hip.so (gfx950) in-tree with hipcc.  No GPU is needed: hipcc cross-compiles.

    python -m tedm_amd.build [--force]

The library carries a digest of the sources it was built from (`tedm_source_digest()`, a generated translation unit): it is rebuilt
when that digest differs from the digest of the sources on disk (or a source is newer than it), and `tedm_amd._lib` refuses to load a
library whose digest does not match the tree it sits in -- a stale binary cannot pass for the current code, here or on a GPU box.
"""
from __future__ import annotations

import hashlib
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


def source_digest() -> str:
    """sha256 over (name, contents) of every file the library is compiled from, in sorted order."""
    h = hashlib.sha256()
    for p in _deps():
        h.update(p.name.encode() + b'\0')
        h.update(p.read_bytes())
        h.update(b'\0')
    return h.hexdigest()


def library_digest(path: Path = LIB):
    """The digest a built library reports, or None (missing library / symbol).  Loading needs no GPU."""
    import ctypes
    if not Path(path).exists():
        return None
    try:
        fn = ctypes.CDLL(str(path)).tedm_source_digest
    except (OSError, AttributeError):
        return None
    fn.restype = ctypes.c_char_p
    return fn().decode()


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    if any(p.stat().st_mtime > t for p in _deps()):
        return True
    return library_digest() != source_digest()


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
    # the digest translation unit (generated: not one of the hashed sources)
    dsrc = OBJ / 'source_digest.cpp'
    dsrc.write_text('extern "C" const char* tedm_source_digest(void) { return "%s"; }\n' % source_digest())
    dobj = OBJ / 'source_digest.o'
    r = subprocess.run([_hipcc(), '-O1', '-fPIC', '-c', '-x', 'c++', str(dsrc), '-o', str(dobj)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for the digest unit:\n{r.stdout}\n{r.stderr}')
    objs.append(dobj)
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(f'built {LIB} ({LIB.stat().st_size} bytes)')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
