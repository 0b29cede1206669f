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
