"""CPU: the bench.py output contract, checked on the newest committed measurement (profiles/*_bench.json = the JSON line the
driver records): required keys, the roofline / cpu_baseline objects, internal consistency of the reported numbers."""
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def newest(pattern):
    files = sorted((ROOT / 'profiles').glob(pattern))
    assert files, pattern
    return json.loads(files[-1].read_text())


def test_headline_line_has_the_contract_fields_and_is_consistent():
    d = newest('r*_bench.json')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'img/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['data'] == 'synthetic' and 'workload' in d['config'] and 'BASELINE configs[1]' in d['config']['workload']
    imgs = d['n_gpus'] * d['config']['images_per_gpu']
    assert abs(d['value'] - imgs / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']          # value = images of all ranks / step time
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'launches', 'avg_us'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 2e-4 and 0 < r['frac'] < 1
    per_launch = r['alg_mb_per_launch'] * 1e6 if r['bound'] == 'hbm' else r['alg_gflop_per_launch'] * 1e9
    unit = 1e9 if r['bound'] == 'hbm' else 1e12
    assert abs(per_launch / (r['avg_us'] * 1e-6) / unit - r['achieved']) < 2e-2 * r['achieved']   # achieved = work per launch / mean duration
    assert r['launches'] * r['avg_us'] * 1e-3 <= d['steps'] * d['ms_per_step']            # the kernel's time fits inside the timed region
    if r['traffic'] is not None and r['bound'] == 'hbm':
        assert r['traffic'] >= 0.9 * r['alg_mb_per_launch'] * 1e6                         # counter traffic cannot be below the algorithmic bytes
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0
    assert d['eps_rel_l2_vs_cpu'] < d['eps_gate'] == 1e-3
    if 'eps_check' in d:                                                                   # round 3: the whole batch, worst case pinned
        assert d['eps_check']['images'] == d['config']['images_per_gpu'] and d['eps_check']['t0_sample_included']
        assert d['eps_check']['worst_sample_rel_l2'] < d['eps_gate']


def test_secondary_modes_carry_workloads_of_their_configs():
    for name, tag in (('extract', 'configs[2]'), ('train', 'configs[3]'), ('tedm', 'configs[4]')):
        d = newest(f'r*_bench_{name}.json')
        assert tag in d['config']['workload'] and d['value'] > 0 and d['vs_baseline'] is None, name
