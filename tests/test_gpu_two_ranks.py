"""bench.py with TWO ranks (python -m torch.distributed.run --nproc-per-node 2), both on the one GPU of the box: RCCL
refuses two ranks per device, so the process group is gloo and the exchanger stages the halo tensors through the host
(SLF_DIST_BACKEND=gloo, SLF_FORCE_DEVICE=0).  Everything else is what the driver's N > 1 runs execute: rank / world
bookkeeping, the global box of the initial state, ring neighbours, barrier + max-over-ranks timing, the gathered
per-rank figures in the JSON line.  Functional, not a performance number."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('mode', [['--scaling', 'weak', '--size', '96'],
                                  ['--scaling', 'strong', '--domain', '128x64x96', '--axis', 'z'],
                                  ['--scaling', 'strong', '--domain', '256x48x40', '--axis', 'x']],
                         ids=['weak_z', 'strong_z', 'strong_x'])
def test_bench_with_two_ranks_on_one_gpu(mode):
    env = dict(os.environ, SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
           '--prewarm_steps', '2', '--repeats', '1', '--no_cpu_baseline', '--no_gpu_state'] + mode
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0, out[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]             # rank 0 only
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['scaling'] == mode[1]
    assert c['rccl_ranks'] == 2 and c['dist_backend'] == 'gloo'
    assert sorted(r['rank'] for r in c['per_rank']) == [0, 1]
    assert all(r['kernel_ms'] > 0 and r['halo_ms'] > 0 for r in c['per_rank'])
    assert set(c['candidates_mlups']) == {'AA', 'AB'}


@pytest.mark.parametrize('axis,pattern,model', [('z', 'AA', 'bgk'), ('z', 'AB', 'mrt'), ('x', 'AA', 'bgk'), ('x', 'AB', 'bgk'),
                                                ('y', 'AB', 'bgk')])
def test_two_processes_equal_one_box(axis, pattern, model):
    """Two OS processes, one slab each, halos through torch.distributed: bit-identical to the undivided box."""
    env = dict(os.environ, SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', '_two_rank_worker.py'), axis, pattern, model]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0 and 'TWO_RANK_PARITY OK backend=gloo world=2' in out, out[-3000:]
