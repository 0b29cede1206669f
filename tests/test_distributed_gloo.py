"""The N > 1 path on the CPU: world_size 2, backend gloo, one subdomain per process, halos through
sailfish_amd.connector.TorchDistConnector -- the same exchange code that runs over RCCL on the GPUs.
The merged result must equal the single-subdomain run bit for bit."""
import os
import socket
import tempfile

import numpy as np
import pytest

from tests import _host
from tests._gloo_worker import controller_worker, worker
from tests._oracle_group import OracleGroup


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = [
    ('ldc_3d', 'LDCSim', 3, 'EqualSubdomainsGeometry3D',
     dict(lat_nx=16, lat_ny=12, lat_nz=12, visc=0.03, access_pattern='AA', subdomains=2, conn_axis='z')),
    ('poiseuille', 'PoiseuilleSim', 2, 'EqualSubdomainsGeometry2D',
     dict(lat_nx=20, lat_ny=24, visc=0.1, horizontal=False, stationary=False, drive='force', wall='fullbb',
          force_implementation='guo', access_pattern='AB', subdomains=2, conn_axis='y')),
]


@pytest.mark.parametrize('case', CASES, ids=['ldc3d_AA_z', 'channel_AB_periodic_y'])
def test_two_ranks_equal_single_subdomain(case):
    import torch.multiprocessing as mp
    steps = 8
    module, sim, dim, geo, cfg = case
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(2, _free_port(), case, steps, d), nprocs=2, join=True)
        parts = [np.load(os.path.join(d, 'rank%d.npz' % r)) for r in range(2)]
    one_cfg = dict(cfg, subdomains=1)
    one = OracleGroup(_host.load_sim_class(module, sim), dim, geo, one_cfg)
    one.run(steps, save_last=True)
    ref_f, ref_rho = one.merged('dist'), one.merged('rho')
    got_f, got_rho = np.zeros_like(ref_f), np.zeros_like(ref_rho)
    for p in parts:
        sl = tuple(slice(int(o), int(o + n)) for o, n in zip(reversed(p['location']), reversed(p['size'])))
        got_f[(slice(None),) + sl] = p['dist']
        got_rho[sl] = p['rho']
    assert np.array_equal(got_f, ref_f, equal_nan=True)
    assert np.array_equal(got_rho, ref_rho, equal_nan=True)


@pytest.mark.parametrize('world', [2, 3])
def test_ring_exchanger(world):
    """The nearest-neighbour ring of bench.py --gpus N: what rank r sends up arrives in the recv_low buffer
    of rank r + 1, what it sends down in the recv_high buffer of rank r - 1 -- including world = 2, where
    both messages travel between the same pair of ranks and are told apart by posting order only."""
    import torch.multiprocessing as mp
    from tests._gloo_worker import ring_worker
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(ring_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        for r in range(world):
            got = np.load(os.path.join(d, 'ring%d.npy' % r))
            down, up = (r - 1) % world, (r + 1) % world
            for step in range(3):
                low0, low1, high0, high1 = got[step]
                assert low0 == low1 == 100.0 * down + 10.0 * step + 1.0      # the lower neighbour's send_up
                assert high0 == high1 == 100.0 * up + 10.0 * step + 2.0      # the upper neighbour's send_down


@pytest.mark.parametrize('single', [False, True], ids=['binary', 'single_component'])
def test_shan_chen_two_ranks(single):
    """Non-local models across processes (config 5 on several GPUs): macro-field exchange + population exchange
    per step over the product's connector; equal to the single-subdomain run bit for bit."""
    import torch.multiprocessing as mp
    from tests import _sc
    from tests._gloo_worker import nn_worker
    from tests._oracle_group import OracleNNGroup
    dim, size, axis, steps = 3, (10, 8, 6), 'z', 6
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(nn_worker, args=(2, _free_port(), dim, size, axis, single, steps, d), nprocs=2, join=True)
        parts = [np.load(os.path.join(d, 'nn%d.npz' % r)) for r in range(2)]
    sim_cls, _ = (_sc.make_single_sim if single else _sc.make_sim)(dim)
    cfg = (_sc.single_config if single else _sc.config)(dim, size, pattern='AA')
    if single:
        cfg.update(G=-1.2, sc_potential='linear')
    cfg.update(subdomains=1, conn_axis=axis)
    one = OracleNNGroup(sim_cls, dim, 'EqualSubdomainsGeometry3D', cfg, single=single)
    one.run(steps)
    ref_f = one.merged((lambda s: s.current()) if single else (lambda s: s.current()[0]))
    ref_rho = one.merged(lambda s: s.rho)
    got_f, got_rho = np.zeros_like(ref_f), np.zeros_like(ref_rho)
    for p in parts:
        sl = tuple(slice(int(o), int(o + n)) for o, n in zip(reversed(p['location']), reversed(p['size'])))
        got_f[(slice(None),) + sl] = p['dist']
        got_rho[sl] = p['rho']
    assert np.array_equal(got_f, ref_f) and np.array_equal(got_rho, ref_rho)


def _merge(parts, ref_f, ref_rho):
    got_f, got_rho = np.zeros_like(ref_f), np.zeros_like(ref_rho)
    for p in parts:
        sl = tuple(slice(int(o), int(o + n)) for o, n in zip(reversed(p['location']), reversed(p['size'])))
        got_f[(slice(None),) + sl] = p['dist']
        got_rho[sl] = p['rho']
    return got_f, got_rho


@pytest.mark.parametrize('case', CASES + [
    ('ldc_3d', 'LDCSim', 3, 'EqualSubdomainsGeometry3D',
     dict(lat_nx=18, lat_ny=10, lat_nz=8, visc=0.03, access_pattern='AB', subdomains=2, conn_axis='x')),
], ids=['ldc3d_AA_z', 'channel_AB_periodic_y', 'ldc3d_AB_x'])
def test_controller_branch_for_world_size_2(case):
    """The real `world > 1` branch of LBSimulationController.run() (controller.py) -- one process per subdomain,
    SubdomainRunner.run(), runner.step(), halo_messages(), TorchDistConnector.exchange(runner) -- on two gloo ranks
    with the CPU test backend.  Merged result == one subdomain, bit for bit."""
    import torch.multiprocessing as mp
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
