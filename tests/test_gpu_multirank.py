"""GPU (one MI355X): rehearsal of the one-process-per-GPU launch that the driver runs at N = 2, 4, 8 -- two ranks started by
torch.distributed.run share the single card (TEDM_DIST_BACKEND=gloo: RCCL refuses two ranks on one device), exercising
bench.py's rank plumbing, the barrier / max-over-ranks timing and, in train mode, the flat-gradient all-reduce."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run_bench(port, *extra):
    env = dict(os.environ, TEDM_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2',
           '--size', '64', '--no-cpu-baseline', *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints exactly one JSON line
    return json.loads(lines[0])


def test_two_ranks_denoise_line(built_lib):
    d = run_bench(29541)
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['unit'] == 'img/s' and d['vs_baseline'] is None
    assert d['value'] > 0 and abs(d['value'] - 2 * 2 * d['steps'] / (d['ms_per_step'] * d['steps'] / 1e3)) < 1e-2 * d['value']
    assert 'roofline' in d and 'cpu_baseline' not in d      # the CPU baseline is an N = 1 leg
    assert d['dist_backend'] == 'gloo' and d['world_size'] == 2     # what torch.distributed itself saw (the driver's runs: nccl = RCCL)


def test_two_ranks_train_line(built_lib):
    """Train mode on 2 ranks: the bucketed all-reduce (4 slices launched from the backward's mark callback on a side stream)
    must leave the same model as ONE all-reduce after the backward: equal loss after 3 optimizer steps."""
    d = run_bench(29542, '--mode', 'train')
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['config']['final_loss'] > 0 and d['config']['grad_allreduce_buckets'] == 4
    # the overlap evidence one N > 1 driver run has to carry: backend / world size as torch.distributed saw
    # them, and per bucket when the backward had produced it, when its all-reduce had finished, GB/s; when the backward finished
    assert d['dist_backend'] == 'gloo' and d['world_size'] == 2
    ar = d['grad_allreduce']
    assert [b['last_group'] for b in ar['buckets']] == ['ups.1', 'ups.0', 'mid_block1', 'time_mlp'] and ar['grad_buckets'] == 4
    assert sum(b['elements'] for b in ar['buckets']) == 36245377
    assert all(b['allreduce_done_ms'] >= b['closed_ms'] > 0 and b['allreduce_gb_per_s'] > 0 for b in ar['buckets'])
    assert ar['backward_done_ms'] >= ar['buckets'][-1]['closed_ms'] and isinstance(ar['hidden_under_backward'], bool)
    d1 = run_bench(29543, '--mode', 'train', '--buckets', '1')
    assert d1['config']['grad_allreduce_buckets'] == 1
    assert abs(d1['config']['final_loss'] - d['config']['final_loss']) <= 1e-6, (d1['config']['final_loss'], d['config']['final_loss'])
