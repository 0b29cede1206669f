"""The controller starts one process per subdomain itself when --gpus names several devices (sailfish_amd/launch.py;
reference master.py:106-117, 242-312): the pure placement logic, and -- with the CPU test backend over gloo -- the
whole path: LBSimulationController.run() in THIS process spawns the ranks, every rank runs the product's
process-per-subdomain branch, the output files merged over the subdomains equal the undivided run."""
import os
import pickle

import numpy as np
import pytest

from sailfish_amd import launch
from tests import _host


def test_cpu_list_and_numa_lookup(tmp_path):
    assert launch.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert launch.parse_cpulist('') == []
    dev = tmp_path / 'bus' / 'pci' / 'devices' / '0000:05:00.0'
    dev.mkdir(parents=True)
    (dev / 'numa_node').write_text('1\n')
    node = tmp_path / 'devices' / 'system' / 'node' / 'node1'
    node.mkdir(parents=True)
    (node / 'cpulist').write_text('64-127,192-255\n')
    assert launch.gpu_numa_node('0000:05:00.0', sysfs=str(tmp_path)) == 1
    assert launch.gpu_numa_node('0000:06:00.0', sysfs=str(tmp_path)) == -1
    assert len(launch.numa_cpus(1, sysfs=str(tmp_path))) == 128
    quota = tmp_path / 'cpu.max'
    quota.write_text('1600000 100000\n')
    assert launch.cpu_quota(str(quota)) == 16
    quota.write_text('max 100000\n')
    assert launch.cpu_quota(str(quota)) is None


def test_ranks_get_disjoint_cores_of_their_gpus_node():
    node_cpus = {0: list(range(0, 64)), 1: list(range(64, 128))}.get
    # 8 GPUs, 4 per socket, every CPU allowed
    sets = launch.cpu_sets([0, 0, 0, 0, 1, 1, 1, 1], range(128), lambda n: node_cpus(n, []))
    assert all(len(s) == 16 for s in sets) and sum(len(set(a) & set(b)) for i, a in enumerate(sets) for b in sets[i + 1:]) == 0
    assert all(c < 64 for s in sets[:4] for c in s) and all(c >= 64 for s in sets[4:] for c in s)
    # a container that only grants 16 CPUs of socket 0: the ranks of socket 1 share what is left, nobody is empty
    sets = launch.cpu_sets([0, 0, 1, 1], range(16), lambda n: node_cpus(n, []), per_rank=4)
    assert all(sets) and all(len(s) <= 8 for s in sets) and set(sets[0]).isdisjoint(sets[1])
    # unknown topology: the allowed CPUs are dealt out evenly
    sets = launch.cpu_sets([-1, -1], range(8), lambda n: [])
    assert sets == [[0, 1, 2, 3], [4, 5, 6, 7]]


def test_gpu_round_robin_and_process_group_backend():
    assert launch.plan_ranks(8, list(range(8))) == (list(range(8)), 'nccl')
    assert launch.plan_ranks(4, [0, 1]) == ([0, 1, 0, 1], 'gloo')          # ranks share GPUs: RCCL refuses that
    assert launch.plan_ranks(2, [0, 0]) == ([0, 0], 'gloo')


def test_config_travels_to_the_ranks_as_plain_values():
    cfg = _host.make_config(3, lat_nx=10)
    d = launch.picklable_config(cfg)
    assert 'logger' not in d and d['lat_nx'] == 10
    pickle.dumps(d)


@pytest.mark.parametrize('pattern,axis,nsub', [('AA', 'z', 2), ('AB', 'x', 2), ('AA', 'x', 3), ('AB', 'x', 8)])
def test_controller_starts_its_own_ranks(pattern, axis, nsub, tmp_path):
    """(nsub = 8: the process count of BASELINE config 4 -- a ring of subdomains whose neighbours are all different
    ranks, the launcher, the rendezvous and the CPU pinning with eight ranks.)"""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    from utils.merge_subdomains import merge_subdomains
    sim_cls = _host.load_sim_class('ldc_3d', 'LDCSim')
    steps = 7
    base = dict(lat_nx=24 if nsub == 8 else 18, lat_ny=10, lat_nz=8, visc=0.03, access_pattern=pattern, conn_axis=axis, max_iters=steps,
                quiet=True, perf_stats_every=0, every=steps, backends='tests._oracle_backend', output_compress=False)
    env = dict((k, os.environ.pop(k, None)) for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'))
    try:
        two = LBSimulationController(sim_cls, geo_mod.EqualSubdomainsGeometry3D,
                                     default_config=dict(base, subdomains=nsub, gpus=[0] * nsub, output=str(tmp_path / 'two')))
        two.run(ignore_cmdline=True)
        assert two.runners == []                      # the subdomains ran in their own processes
        one = LBSimulationController(sim_cls, geo_mod.EqualSubdomainsGeometry3D,
                                     default_config=dict(base, subdomains=1, gpus=[0], output=str(tmp_path / 'one')))
        one.run(ignore_cmdline=True)
    finally:
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v
    digits = len(str(steps))
    got = merge_subdomains(str(tmp_path / 'two'), digits, steps, save=False)
    ref = merge_subdomains(str(tmp_path / 'one'), digits, steps, save=False)
    assert set(got) == set(ref) and 'rho' in ref
    for name in ref:
        assert np.array_equal(got[name], ref[name], equal_nan=True), name


@pytest.mark.parametrize('pattern,axis,nsub', [('AA', 'z', 2), ('AB', 'x', 3), ('AA', 'y', 2)])
def test_same_process_group_steps_as_one_program(pattern, axis, nsub, tmp_path):
    """controller.LocalGroup (several subdomains in THIS process, one --gpus entry) with the CPU test backend: the group's
    step -- fronts of all runners, copies between their halo buffers, backs -- gives the undivided run bit for bit."""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    from utils.merge_subdomains import merge_subdomains
    sim_cls = _host.load_sim_class('ldc_3d', 'LDCSim')
    steps = 7
    base = dict(lat_nx=18, lat_ny=12, lat_nz=10, visc=0.03, access_pattern=pattern, conn_axis=axis, max_iters=steps,
                quiet=True, perf_stats_every=0, every=steps, backends='tests._oracle_backend', output_compress=False, gpus=[0])
    for name, n in (('many', nsub), ('one', 1)):
        ctrl = LBSimulationController(sim_cls, geo_mod.EqualSubdomainsGeometry3D,
                                      default_config=dict(base, subdomains=n, output=str(tmp_path / name)))
        ctrl.run(ignore_cmdline=True)
        assert len(ctrl.runners) == n
    got = merge_subdomains(str(tmp_path / 'many'), 1, steps, save=False)
    ref = merge_subdomains(str(tmp_path / 'one'), 1, steps, save=False)
    for name in ref:
        assert np.array_equal(got[name], ref[name], equal_nan=True), name


def test_a_rank_that_dies_ends_the_run_instead_of_hanging_it(tmp_path):
    """One subdomain process fails during set-up: the controller ends the others and raises (it does not wait for ever
    for rank 0, which sits in an exchange with the dead rank)."""
    import time
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    from tests._failing_sim import FailingSim
    env = dict((k, os.environ.pop(k, None)) for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'))
    try:
        ctrl = LBSimulationController(FailingSim, geo_mod.EqualSubdomainsGeometry3D,
                                      default_config=dict(lat_nx=18, lat_ny=10, lat_nz=8, visc=0.03, access_pattern='AB',
                                                          conn_axis='x', max_iters=50, quiet=True, perf_stats_every=0,
                                                          backends='tests._oracle_backend', subdomains=2, gpus=[0, 0]))
        t0 = time.time()
        with pytest.raises(RuntimeError, match='subdomain processes failed'):
            ctrl.run(ignore_cmdline=True)
        assert time.time() - t0 < 120
    finally:
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v
