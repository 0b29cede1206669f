"""The bench.py output contract (one JSON line), checked on the line committed from the last GPU run of the
round, and the argument handling that can be checked without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_every_contract_field():
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r01', 'bench_r01_last.json')))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'MLUPS' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic'
    assert d['vs_baseline'] is None and d['dtype'] == 'f32' and 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    # achieved = algorithmic bytes per launch / kernel time; value = updates / wall time
    assert abs(r['achieved'] - 512 ** 3 * r['bytes_per_update'] / (r['kernel_ms'] * 1e-3) / 1e9) < 1.0
    assert abs(d['value'] - 512 ** 3 / (d['ms_per_step'] * 1e-3) * 1e-6) / d['value'] < 1e-3
    assert r['traffic'] is None or 0.9 < r['traffic'] / (512 ** 3 * 152) < 1.2
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'MLUPS' and c['cores'] >= 1 and 'oracle' in c['sample']


def test_bench_refuses_to_run_without_a_gpu_or_with_inconsistent_ranks():
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    assert res.returncode != 0 and b'torch.distributed.run' in res.stdout
