"""bench.py with TWO ranks (python -m torch.distributed.run --nproc-per-node 2), both on the one GPU of the box: RCCL
refuses two ranks per device, so the process group is gloo (SLF_DIST_BACKEND=gloo, SLF_FORCE_DEVICE=0); the halos travel
through the peer transport -- the neighbour's receive buffers mapped into this process, ordered by progress counters
(sailfish_amd/peer.py; round 6) -- or, with SLF_HALO_TRANSPORT=torch, staged through the host by the exchanger.  Everything else is what the driver's N > 1 runs execute: rank / world
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


@pytest.mark.parametrize('transport', ['peer', 'torch'])
@pytest.mark.parametrize('mode', [['--scaling', 'weak', '--size', '96'],
                                  ['--scaling', 'strong', '--domain', '128x64x96', '--axis', 'z'],
                                  ['--scaling', 'strong', '--domain', '256x48x40', '--axis', 'x']],
                         ids=['weak_z', 'strong_z', 'strong_x'])
def test_bench_with_two_ranks_on_one_gpu(mode, transport):
    if transport == 'torch' and mode[1] == 'weak':
        pytest.skip('host staging is covered by the two strong-scaling cases')
    env = dict(os.environ, SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0',
               SLF_HALO_TRANSPORT=transport, GPU_MAX_HW_QUEUES='2')
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
    assert c['rccl_ranks'] == 0 and c['world_size'] == 2 and c['dist_backend'] == 'gloo'      # gloo: nothing went over RCCL
    assert c['halo_transport'].startswith('peer:' if transport == 'peer' else 'torch.distributed'), c['halo_transport']
    assert c['validated'] is True, c['validation']
    assert sorted(r['rank'] for r in c['per_rank']) == [0, 1]
    assert all(r['kernel_ms'] > 0 and r['halo_ms'] > 0 for r in c['per_rank'])
    assert set(c['candidates_mlups']) == {'AA', 'AB'}


def test_bench_launches_two_ranks_itself():
    """`python bench.py --gpus 2 ...` with NO launcher around it (the shape of the driver's N = 1 command with another
    N): bench.py starts its ranks itself, rank 0 prints the one JSON line, the final state is validated."""
    env = dict(os.environ, SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2', '--size', '96',
           '--prewarm_steps', '2', '--repeats', '1', '--no_cpu_baseline', '--no_gpu_state']
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0, out[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == 2 and c['rccl_ranks'] == 0 and c['world_size'] == 2 and sorted(r['rank'] for r in c['per_rank']) == [0, 1]
    assert c['validated'] is True, c['validation']
    assert all(v['populations_bit_identical'] for v in c['validation'].values())
    assert all(v['undivided_box']['populations_bit_identical'] and v['undivided_box']['slabs'] == 2 for v in c['validation'].values())


def test_results_of_the_peer_transport_that_fail_validation_are_not_reported():
    """The peer transport has never run between two GPUs (single-GPU boxes): if the seam layers it produced did not match
    the oracle, bench.py repeats the measurement over RCCL / torch.distributed and reports THAT, with the rejected figures
    beside it (here the rejection is forced: SLF_BENCH_TEST_REJECT_PEER=1)."""
    env = dict(os.environ, SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0',
               SLF_BENCH_TEST_REJECT_PEER='1', GPU_MAX_HW_QUEUES='2')
    env.pop('SLF_HALO_TRANSPORT', None)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2', '--scaling', 'strong',
           '--domain', '256x48x40', '--axis', 'x', '--prewarm_steps', '2', '--repeats', '1', '--no_cpu_baseline', '--no_gpu_state',
           '--access_pattern', 'AA']
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0, out[-3000:]
    d = json.loads([ln for ln in out.splitlines() if ln.startswith('{"metric"')][-1])
    c = d['config']
    assert c['halo_transport'].startswith('torch.distributed') and c['validated'] is True
    rej = c['peer_transport_rejected']
    assert rej['halo_transport'] == 'peer' and rej['mlups']['AA'] > 0 and rej['validation']['AA']['test_forced_rejection']


@pytest.mark.parametrize('how', ['stall', 'exit'])
def test_a_rank_that_stops_ends_the_bench_with_an_error_line(how):
    """One of two ranks stops advancing (or leaves) after the rendezvous: instead of hanging until the launcher's limit,
    rank 0 prints ONE line with "error", the phase and every rank's last state, and the run returns a non-zero status
    within the deadline (sailfish_amd/watchdog.py; reference master.py:268-312)."""
    import time
    env = dict(os.environ, SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0',
               SLF_BENCH_TEST_STALL='1:first_exchange' + (':exit' if how == 'exit' else ''), SLF_DEADLINE_SCALE='0.1',
               SLF_DEADLINE_FIRST_EXCHANGE='8', SLF_DEADLINE_START='300', SLF_PEER_TIMEOUT_S='60', GPU_MAX_HW_QUEUES='2')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2', '--size', '64',
           '--prewarm_steps', '2', '--repeats', '1', '--no_cpu_baseline', '--no_gpu_state', '--access_pattern', 'AA']
    t0 = time.time()
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=300)
    took = time.time() - t0
    out = res.stdout.decode(errors='replace')
    assert res.returncode != 0, out[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    d = json.loads(lines[0])
    assert d['value'] is None and d['error'] and len(d['ranks']) == 2 and took < 150, (took, d)
    states = dict((s['rank'], s) for s in d['ranks'])
    if how == 'stall':
        assert states[1]['phase'] == 'first_exchange' and 'first_exchange' in d['error']
    else:
        assert 'rank 1' in d['error'] or 'SIGTERM' in d['error']


@pytest.mark.parametrize('axis,pattern,model,transport', [('z', 'AA', 'bgk', 'peer'), ('z', 'AB', 'mrt', 'peer'), ('x', 'AA', 'bgk', 'peer'),
                                                          ('x', 'AB', 'bgk', 'peer'), ('y', 'AB', 'bgk', 'peer'),
                                                          ('z', 'AA', 'bgk', 'torch'), ('x', 'AA', 'bgk', 'torch')])
def test_two_processes_equal_one_box(axis, pattern, model, transport):
    """Two OS processes, one slab each, halos written straight into the neighbour's buffers (peer) or staged through
    torch.distributed (torch): bit-identical to the undivided box."""
    env = dict(os.environ, SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0',
               SLF_HALO_TRANSPORT=transport, GPU_MAX_HW_QUEUES='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', '_two_rank_worker.py'), axis, pattern, model]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0 and 'TWO_RANK_PARITY OK backend=gloo world=2' in out, out[-3000:]


def _controller_cases():
    from tests.test_distributed_gloo import CASES
    return CASES + [('ldc_3d', 'LDCSim', 3, 'EqualSubdomainsGeometry3D',
                     dict(lat_nx=18, lat_ny=10, lat_nz=8, visc=0.03, access_pattern='AB', subdomains=2, conn_axis='x'))]


@pytest.mark.parametrize('case', _controller_cases(), ids=['ldc3d_AA_z', 'channel_AB_periodic_y', 'ldc3d_AB_x'])
def test_controller_branch_two_processes_on_the_gpu(case, monkeypatch):
    """The product's process-per-subdomain branch (LBSimulationController.run() under WORLD_SIZE > 1 ->
    SubdomainRunner.run() -> step() -> halo_messages() -> TorchDistConnector.exchange(runner)) with the HIP backend in
    two OS processes on the one GPU of the box (gloo group, halo tensors staged through the host).  The merged
    populations and densities equal the single-subdomain oracle run bit for bit."""
    import tempfile
    import numpy as np
    import torch.multiprocessing as mp
    from tests import _host
    from tests._gloo_worker import controller_worker
    from tests._oracle_group import OracleGroup
    from tests.test_distributed_gloo import _merge
    for k, v in (('SLF_DIST_BACKEND', 'gloo'), ('SLF_FORCE_DEVICE', '0'), ('SLF_TEST_CONTROLLER_ON_GPU', '1')):
        monkeypatch.setenv(k, v)
    steps = 7
    module, sim, dim, geo, cfg = case
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(controller_worker, args=(2, _free_port(), case, steps, d), nprocs=2, join=True)
        parts = [np.load(os.path.join(d, 'rank%d.npz' % r)) for r in range(2)]
    one = OracleGroup(_host.load_sim_class(module, sim), dim, geo, dict(cfg, subdomains=1))
    one.run(steps, save_last=True)
    ref_f, ref_rho = one.merged('dist'), one.merged('rho')
    got_f, got_rho = _merge(parts, ref_f, ref_rho)
    assert np.array_equal(got_f, ref_f, equal_nan=True)
    assert np.array_equal(got_rho, ref_rho, equal_nan=True)


@pytest.mark.parametrize('pattern,axis,nsub', [('AA', 'x', 2), ('AB', 'z', 2), ('AA', 'x', 8), ('AB', 'y', 8)])
def test_example_starts_its_own_ranks(pattern, axis, nsub, tmp_path):
    """`python examples/ldc_3d.py --subdomains=2 --gpus 0 0` with NO launcher: the controller starts one process per
    subdomain itself (sailfish_amd/launch.py; reference master.py:242-312) -- two (eight: the process count of BASELINE
    config 4, neighbours all different) ranks on the one GPU of the box, hence a gloo group -- and the merged output
    equals the single-subdomain run of the same script."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from utils.merge_subdomains import merge_subdomains
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'SLF_DIST_BACKEND', 'SLF_FORCE_DEVICE'):
        env.pop(k, None)
    steps = 9
    size = {'x': (48, 12, 10), 'y': (14, 40, 10), 'z': (40, 12, 10)}[axis] if nsub == 8 else (40, 12, 10)
    common = ['--lat_nx=%d' % size[0], '--lat_ny=%d' % size[1], '--lat_nz=%d' % size[2], '--visc=0.03', '--max_iters=%d' % steps,
              '--every=%d' % steps, '--access_pattern=' + pattern, '--conn_axis=' + axis, '--quiet', '--nooutput_compress',
              '--perf_stats_every=0']
    for name, extra in (('two', ['--subdomains=%d' % nsub, '--gpus'] + ['0'] * nsub), ('one', ['--subdomains=1', '--gpus', '0'])):
        cmd = [sys.executable, os.path.join(ROOT, 'examples', 'ldc_3d.py'), '--output=' + str(tmp_path / name)] + common + extra
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
        assert res.returncode == 0, res.stdout.decode(errors='replace')[-3000:]
    got = merge_subdomains(str(tmp_path / 'two'), 1, steps, save=False)
    ref = merge_subdomains(str(tmp_path / 'one'), 1, steps, save=False)
    assert set(got) == set(ref) and 'rho' in ref and 'v' in ref
    for name in ref:
        assert np.array_equal(got[name], ref[name], equal_nan=True), name


@pytest.mark.parametrize('axis,nsub,transport', [('x', 2, 'auto'), ('x', 3, 'torch'), ('z', 2, 'auto'), ('y', 4, 'auto')])
def test_shan_chen_mixture_one_process_per_subdomain(axis, nsub, transport, tmp_path):
    """The binary Shan-Chen model with one PROCESS per subdomain on the one GPU of the box: the density planes that the
    force kernel reads across the seam and the distributions both travel through the device-side peer transport
    (connector.PeerConnector, receive buffers by step parity), and the merged fields equal the single-subdomain run of
    the same script bit for bit (reference lb_binary.py:393-433 macro exchange + lb_base.py:232-256)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from utils.merge_subdomains import merge_subdomains
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'SLF_DIST_BACKEND', 'SLF_FORCE_DEVICE'):
        env.pop(k, None)
    env['SLF_HALO_TRANSPORT'] = transport        # torch: the planes as torch.distributed messages (staged through the host)
    steps = 10
    common = ['--max_iters=%d' % steps, '--every=%d' % steps, '--conn_axis=' + axis, '--verbose', '--nooutput_compress',
              '--perf_stats_every=0']
    for name, extra in (('many', ['--subdomains=%d' % nsub, '--gpus'] + ['0'] * nsub), ('one', ['--subdomains=1', '--gpus', '0'])):
        cmd = [sys.executable, os.path.join(ROOT, 'tests', '_sc_ranks_script.py'), '--output=' + str(tmp_path / name)] + common + extra
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
        text = res.stdout.decode(errors='replace')
        assert res.returncode == 0, text[-3000:]
        # slabs along x: no ghost columns, the kernels write / read dense planes in the neighbour's memory (xface.NNPlanes)
        assert ('Shan-Chen model over x-face planes' in text) == (axis == 'x' and name == 'many'), text[-3000:]
    got = merge_subdomains(str(tmp_path / 'many'), 1, steps, save=False)
    ref = merge_subdomains(str(tmp_path / 'one'), 1, steps, save=False)
    assert set(got) == set(ref) and 'rho' in ref and 'phi' in ref
    for name in ref:
        assert np.array_equal(got[name], ref[name], equal_nan=True), name
    assert np.ptp(ref['phi']) > 1e-3


# ---- the same paths over RCCL: run by themselves on any box with at least two GPUs (an 8-GPU node exercises the real
# ---- transport -- DirectRccl communicator of two ranks, step plans with RCCL batches to another device -- without anyone
# ---- asking); skipped on the 1-GPU boxes of the build pool.
def _gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        return 0


needs_two_gpus = pytest.mark.skipif(_gpus() < 2, reason='needs two GPUs (RCCL refuses two ranks per device)')


def _rccl_env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'SLF_DIST_BACKEND', 'SLF_FORCE_DEVICE'):
        env.pop(k, None)
    return env


@needs_two_gpus
@pytest.mark.parametrize('mode', [['--scaling', 'weak', '--size', '96'],
                                  ['--scaling', 'strong', '--domain', '128x64x96', '--axis', 'z'],
                                  ['--scaling', 'strong', '--domain', '256x48x40', '--axis', 'x']],
                         ids=['weak_z', 'strong_z', 'strong_x'])
def test_bench_with_two_rccl_ranks(mode):
    """`python bench.py --gpus 2` on two GPUs: RCCL process group, halos through the C-ABI communicator inside the step
    plans, seam layers validated on both ranks."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2', '--prewarm_steps', '2',
           '--repeats', '1', '--no_cpu_baseline', '--no_gpu_state', '--min_seconds', '0.05'] + mode
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=_rccl_env(), timeout=900)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0, out[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    c = json.loads(lines[0])['config']
    assert c['rccl_ranks'] == 2 and c['dist_backend'] == 'nccl' and 'C ABI' in c['halo_transport']
    assert sorted(r['device'] for r in c['per_rank']) == [0, 1] and all(r['step_plans'] for r in c['per_rank'])
    assert c['validated'] is True and all(v['populations_bit_identical'] and v['ranks_checked'] == 2 for v in c['validation'].values())
    assert all(v['undivided_box']['populations_bit_identical'] for v in c['validation'].values())


@needs_two_gpus
@pytest.mark.parametrize('axis,pattern,model', [('z', 'AA', 'bgk'), ('x', 'AA', 'bgk'), ('x', 'AB', 'mrt'), ('y', 'AB', 'bgk')])
def test_two_rccl_ranks_equal_one_box(axis, pattern, model):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', '_two_rank_worker.py'), axis, pattern, model]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=_rccl_env(), timeout=900)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0 and 'TWO_RANK_PARITY OK backend=nccl world=2' in out, out[-3000:]


@needs_two_gpus
@pytest.mark.parametrize('pattern,axis', [('AA', 'x'), ('AB', 'z')])
def test_example_starts_its_own_rccl_ranks(pattern, axis, tmp_path):
    """`python examples/ldc_3d.py --subdomains=2 --gpus 0 1`: the controller starts one process per GPU, the runners'
    step plans carry the RCCL batches; merged output == the single-subdomain run."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from utils.merge_subdomains import merge_subdomains
    steps = 9
    common = ['--lat_nx=40', '--lat_ny=12', '--lat_nz=10', '--visc=0.03', '--max_iters=%d' % steps, '--every=%d' % steps,
              '--access_pattern=' + pattern, '--conn_axis=' + axis, '--quiet', '--nooutput_compress', '--perf_stats_every=0']
    for name, extra in (('two', ['--subdomains=2', '--gpus', '0', '1']), ('one', ['--subdomains=1', '--gpus', '0'])):
        cmd = [sys.executable, os.path.join(ROOT, 'examples', 'ldc_3d.py'), '--output=' + str(tmp_path / name)] + common + extra
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=_rccl_env(), timeout=900)
        assert res.returncode == 0, res.stdout.decode(errors='replace')[-3000:]
    got = merge_subdomains(str(tmp_path / 'two'), 1, steps, save=False)
    ref = merge_subdomains(str(tmp_path / 'one'), 1, steps, save=False)
    for name in ref:
        assert np.array_equal(got[name], ref[name], equal_nan=True), name


@pytest.mark.parametrize('mode', [['--scaling', 'weak', '--size', '64'],
                                  ['--scaling', 'strong', '--domain', '256x24x40', '--axis', 'x']], ids=['weak_z', 'strong_x'])
def test_bench_with_four_ranks_on_one_gpu(mode):
    """Four ranks (gloo, one GPU): a ring whose up and down neighbours are DIFFERENT ranks -- with two ranks they
    coincide -- through the slab exchange, the seam validation (every rank's windows take layers of both neighbours)
    and the gathered per-rank figures."""
    env = dict(os.environ, SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '4', '--warmup', '2', '--prewarm_steps', '2',
           '--repeats', '1', '--no_cpu_baseline', '--no_gpu_state', '--min_seconds', '0.02', '--halo_timing_steps', '4'] + mode
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=900)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0, out[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == 4 and c['rccl_ranks'] == 0 and c['world_size'] == 4 and sorted(r['rank'] for r in c['per_rank']) == [0, 1, 2, 3]
    assert c['validated'] is True, c['validation']
    assert all(v['populations_bit_identical'] and v['ranks_checked'] == 4 for v in c['validation'].values())
    assert all(v['undivided_box']['populations_bit_identical'] and v['undivided_box']['slabs'] == 4 for v in c['validation'].values())
