"""Every script under examples/ (and the weak-scaling harness) runs from the command line on the GPU, the way
a user starts it: options parse, the run completes, output files appear (reference tests/run_examples.sh)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    ('examples/ldc_2d.py', ['--lat_nx=64', '--lat_ny=48', '--visc=0.05']),
    ('examples/ldc_3d.py', ['--lat_nx=32', '--lat_ny=24', '--lat_nz=20', '--visc=0.05', '--model=mrt',
                            '--access_pattern=AA']),
    ('examples/poiseuille.py', ['--lat_nx=32', '--lat_ny=48', '--visc=0.1', '--drive=pressure']),
    ('examples/poiseuille_3d.py', ['--lat_nx=24', '--lat_ny=24', '--lat_nz=32', '--visc=0.05', '--subdomains=2',
                                   '--conn_axis=z']),
    ('examples/sc_phase_separation.py', ['--lat_nx=64', '--lat_ny=64']),
    ('examples/binary_fluid/sc_separation_3d.py', ['--lat_nx=32', '--lat_ny=24', '--lat_nz=16',
                                                   '--force_implementation=edm']),
    ('examples/external_geometry.py', ['--lat_nx=64', '--lat_ny=21', '--lat_nz=21', '--node_addressing=indirect']),
]


@pytest.mark.parametrize('script,opts', CASES, ids=[c[0].split('/')[-1] for c in CASES])
def test_example_runs_from_the_command_line(script, opts, tmp_path):
    out = str(tmp_path / 'run')
    cmd = [sys.executable, os.path.join(ROOT, script), '--max_iters=40', '--every=20', '--output=' + out,
           '--quiet'] + opts
    res = subprocess.run(cmd, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert res.returncode == 0, res.stdout.decode()[-2000:]
    files = sorted(f for f in os.listdir(str(tmp_path)) if f.startswith('run.') and f.endswith('.npz'))
    assert files, os.listdir(str(tmp_path))
    last = np.load(os.path.join(str(tmp_path), files[-1]))
    assert 'rho' in last.files and 'v' in last.files
    rho = last['rho']
    assert np.isfinite(rho[~np.isnan(rho)]).all() and (~np.isnan(rho)).any()


def test_weak_scaling_harness(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, 'benchmark', 'scaling', 'weak_single_3d.py'), '--num_blocks', '2',
           '--edge', '48']
    res = subprocess.run(cmd, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert res.returncode == 0, res.stdout.decode()[-2000:]
    assert b'weak_3d_single blocks=2' in res.stdout
    assert os.path.exists(os.path.join(str(tmp_path), 'weak_3d_single_mlups_2'))


@pytest.mark.parametrize('script,tag,edge', [('weak_single_3d.py', 'weak_3d_single', 48), ('weak_binary_3d.py', 'weak_3d_binary', 32)])
def test_weak_scaling_harness_starts_its_own_ranks(script, tag, edge, tmp_path):
    """--gpus on the harness's command line reaches the controller, which starts one process per block (two ranks on the
    one GPU here, gloo); the timing summary of all ranks comes back to the script."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'SLF_DIST_BACKEND', 'SLF_FORCE_DEVICE'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'benchmark', 'scaling', script), '--num_blocks', '2', '--edge', str(edge),
           '--gpus', '0', '0']
    res = subprocess.run(cmd, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=900)
    assert res.returncode == 0, res.stdout.decode(errors='replace')[-3000:]
    assert ('%s blocks=2' % tag).encode() in res.stdout
    with open(os.path.join(str(tmp_path), '%s_mlups_2' % tag)) as f:
        eff, comp = [float(x) for x in f.read().split()]
    assert eff > 0 and comp > 0
    with open(os.path.join(str(tmp_path), '%s_2' % tag)) as f:
        assert f.read().count('TimingInfo') == 2                # one per block
