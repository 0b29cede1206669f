"""Parity at the sizes BASELINE.json quotes, with production placement of the distribution arrays (16 physical
chunks spread over HBM, sailfish_amd/placement.py) -- what the reference does with tests/gpu/access_pattern.sh:12-29
and regtest/subdomains/3d_ldc.py:79-95 at real sizes.

  config 2  D3Q19 BGK periodic 256^3: full field, AA and AB, 6 steps, against the blocked OpenMP twin
            (oracle/lbm_fast.c, itself bit-identical to the table-driven oracle: tests/test_cpu_twin.py);
  headline  D3Q19 BGK periodic 512^3: sampled z-planes (first / last plane = the in-sweep wrap, planes holding a
            boundary between two physical chunks) through 7-plane oracle windows (oracle/window.py,
            tests/test_window_oracle.py);
  config 3  D3Q19 MRT lid-driven cavity 512^3 with the node map: sampled planes incl. the wall planes, the lid rows
            (y = ny of every plane) and chunk-boundary planes;
  config 4  one 128 x 512 x 512 x-slab pair (2 of the 8 subdomains of 1024 x 512 x 512) through the x-face buffers
            against the undivided 256 x 512 x 512 box, which is checked against the oracle windows itself;
  config 4  as stated: all EIGHT subdomains of 1024 x 512 x 512 (x-slabs 128 x 512 x 512, z-slabs 1024 x 512 x 64), one
            process each on the one GPU, seam windows on every rank + whole planes against the undivided box;
  config 5  binary Shan-Chen 256^3 through the host stack, 2 steps, full field against the oracle twin.

Populations must be bit-identical (same IEEE operation order, FMA contraction off); rho / u within the north-star
tolerance 1e-6.
"""
import os

import numpy as np
import pytest

from oracle import window
from sailfish_amd import sym
from sailfish_amd.box import BoxSim, make_box_desc
from tests import _geometry as geo
from tests._oracle_box import OracleBox, synthetic_fields

pytestmark = pytest.mark.gpu

RTOL = 1e-6


@pytest.fixture(scope='module')
def backend():
    from sailfish_amd.backend_hip import HIPBackend

    class Opt(object):
        pass
    return HIPBackend(Opt(), 0)


def _sample_planes(sim, extra=()):
    nz = sim.desc.lat_nz - 2
    zs = [1, 2, nz // 2, nz - 1, nz] + list(extra)
    zs += window.chunk_boundary_planes(sim.placed, sim.desc, sim.stride) if sim.placed else []
    out = []
    for z in zs:
        if z not in out:
            out.append(z)
    return out


def _check_planes(backend, sim, zs, warm=4):
    """Advance `warm` steps, seed windows from the device, two more steps on both sides, compare."""
    for _ in range(warm):
        sim.step()
    sim.sync()
    chk = window.PlaneCheck(backend, sim.desc, sim.node_map, zs, sim.gpu_dist, sim.stride,
                            [sim.gpu_rho] + list(sim.gpu_v))
    chk.seed(sim.iteration)
    sim.run(2, save_last=True)
    sim.sync()
    chk.advance(2, save_last=True)
    return chk.compare()


@pytest.mark.parametrize('pattern', ['AA', 'AB'])
def test_config2_box_256_full_field(backend, pattern):
    """BASELINE config 2: every population of the 256^3 box, 6 steps."""
    from oracle import cpu_twin
    size = (256, 256, 256)
    desc = make_box_desc(sym.D3Q19, size, precision='single', access_pattern=pattern, visc=0.01,
                         periodic_fused=[1, 1, 1])
    rho, v = synthetic_fields(size, 3)
    g = BoxSim(backend, desc, periodic=(True, True, True))
    assert g.placed, 'production placement expected at this size'
    g.set_fields(rho, v)
    g.initial_conditions()
    o = OracleBox(make_box_desc(sym.D3Q19, size, precision='single', access_pattern='AA', visc=0.01,
                                periodic_fused=[1, 1, 1]), periodic=(True, True, True))
    o.set_fields(rho, v)
    o.initial_conditions()
    assert np.array_equal(g.real_view(g.get_dist()), o.real_view(o.dist[0])), 'initial state differs'
    twin = cpu_twin.FastBox('D3Q19', size, 0.01, 'single')
    twin.set_dist(o.dist[0])
    fields = [np.full(o.shape, np.inf, dtype=np.float32) for _ in range(4)]
    twin.run(5)
    twin.run(1, fields=fields)
    g.run(6, save_last=True)
    g_rho, g_v = g.fetch_fields()
    assert np.array_equal(g.real_view(g.get_dist()), o.real_view(twin.dist)), 'populations differ after 6 steps'
    assert np.max(np.abs(g.real_view(g_rho) - o.real_view(fields[0]))) < RTOL
    for d in range(3):
        assert np.max(np.abs(g.real_view(g_v[d]) - o.real_view(fields[1 + d]))) < RTOL * 0.05
    g.release()


@pytest.mark.parametrize('pattern', ['AA', 'AB'])
def test_headline_box_512_planes(backend, pattern):
    """The bench.py configuration itself: 512^3 periodic BGK with placed arrays (544-wide rows, 143.7 M node
    indices, chunk boundaries inside the arrays)."""
    size = (512, 512, 512)
    desc = make_box_desc(sym.D3Q19, size, precision='single', access_pattern=pattern, visc=1.0 / 6.0,
                         periodic_fused=[1, 1, 1])
    g = BoxSim(backend, desc, periodic=(True, True, True))
    assert g.placed and g.placement_info['parts'] == 16
    rho, v = synthetic_fields(size, 3, dtype=np.float32)
    g.set_fields(rho, v)
    g.initial_conditions()
    zs = _sample_planes(g)
    assert len(zs) >= 8, zs
    res = _check_planes(backend, g, zs)
    g.release()
    assert res['dist_exact'], res
    assert res['rho_err'] < RTOL and res['v_abs_err'] < RTOL * 0.05, res


@pytest.mark.parametrize('model,pattern', [('mrt', 'AA'), ('bgk', 'AB')])
def test_config3_cavity_512_planes(backend, model, pattern):
    """BASELINE config 3 (MRT lid-driven cavity 512^3, in place) and the same geometry with the reference's default
    two-copy pattern: node-map kernels, walls, the regularized-velocity lid, at full size."""
    size = (512, 512, 512)
    desc = make_box_desc(sym.D3Q19, size, model=model, precision='single', access_pattern=pattern,
                         visc=(size[0] - 2) * 0.05 / 400.0, fluid_only=False, type_kind=geo.TYPE_KIND,
                         nt_bits=geo.NT_BITS, node_params=[0.05, 0.0, 0.0])
    nmap = geo.cavity_3d(desc)
    g = BoxSim(backend, desc, periodic=(False, False, False), node_map=nmap)
    assert g.placed
    shape = tuple(reversed(size))
    g.set_fields(np.ones(shape, dtype=np.float32), [np.zeros(shape, dtype=np.float32) for _ in range(3)])
    g.initial_conditions()
    zs = _sample_planes(g, extra=(3, 509))
    res = _check_planes(backend, g, zs, warm=6)
    g.release()
    assert res['dist_exact'], res
    assert res['rho_err'] < RTOL and res['v_abs_err'] < RTOL * 0.05, res


@pytest.mark.parametrize('pattern', ['AA', 'AB'])
def test_config4_x_slab_pair_full_size(pattern):
    """Two 128 x 512 x 512 x-slabs (the subdomain shape of BASELINE config 4) exchanging through the x-face
    buffers == the undivided 256 x 512 x 512 box on sampled planes, bit for bit; the undivided box against the
    oracle windows."""
    import torch
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.slab import SlabSim
    from tests.test_gpu_slab import Loopback, step_pair

    class Opt(object):
        pass
    n = (128, 512, 512)
    lb = Loopback()
    sims = [SlabSim(HIPBackend(Opt(), 0), sym.D3Q19, n, rank=r, world=2, model='bgk', access_pattern=pattern,
                    visc=0.02, exchanger=lb.bind(r), axis='x') for r in range(2)]
    assert all(s.xface is not None and s.placed for s in sims)
    for s in sims:
        s.init_synthetic(seed=5)
    b1 = HIPBackend(Opt(), 0)
    one = SlabSim(b1, sym.D3Q19, (256, 512, 512), rank=0, world=1, model='bgk', access_pattern=pattern, visc=0.02)
    rho = np.concatenate([s.real_view(s.rho) for s in sims], axis=2)
    v = [np.concatenate([s.real_view(s.v[d]) for s in sims], axis=2) for d in range(3)]
    one.set_fields(rho, v)
    one.initial_conditions()
    steps = 6
    for _ in range(steps):
        step_pair(sims, lb)
        one.step()
    torch.cuda.synchronize()
    for s in sims:
        s.materialise_faces()
    one.sync()
    zs = [1, 2, 100, 255, 256, 257, 400, 511, 512]
    for z in zs:
        for q in range(19):
            ref = _plane(b1, one, q, z)[1:513, 1:257]
            got = np.concatenate([_plane(s.backend, s, q, z)[1:513, 1:129] for s in sims], axis=1)
            assert np.array_equal(got, ref), (z, q)
    # ... and the undivided box against the oracle
    chk = window.PlaneCheck(b1, one.desc, None, [1, 256, 512], one.gpu_dist, one.stride)
    chk.seed(one.iteration)
    one.run(2, save_last=False)
    one.sync()
    chk.advance(2, save_last=False)
    res = chk.compare(fields=False)
    for s in sims + [one]:
        s.release()
    assert res['dist_exact'], res


@pytest.mark.parametrize('axis,transport', [('x', 'peer'), ('z', 'peer'), ('x', 'torch')])
def test_config4_eight_subdomains_full_extent(axis, transport):
    """BASELINE config 4 as stated: D3Q19 BGK 1024 x 512 x 512 cut into EIGHT subdomains (reference geo.py:100-135
    EqualSubdomainsGeometry3D; x = its default axis: 128 x 512 x 512 each, z: 1024 x 512 x 64), one process per subdomain
    -- a ring whose neighbours are all different ranks, with the wrap 7 -> 0 -- at full size.  The box of the build pool
    has ONE GPU, so the eight ranks share it (gloo group for the rendezvous, `rccl_ranks` 0).  transport = peer (round 6,
    the default wherever processes can map each other's memory -- the eight GPUs of a node, or one GPU shared): the sweep's
    edge lanes / the pack kernels store into the neighbouring PROCESS's receive buffers, ordered by progress counters
    (sailfish_amd/peer.py); torch: halo buffers staged through the host by torch.distributed, as in round 5.  After the timed steps every
    rank checks its seam layers through windows that reach into both neighbours (window.SeamCheck) and rank 0 checks
    whole planes of the merged slabs against oracle windows of the UNDIVIDED 1024 x 512 x 512 box (window.GlobalCheck),
    populations bit for bit, both access patterns."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0',
               SLF_HALO_TRANSPORT=transport)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--scaling', 'strong', '--domain', '1024x512x512',
           '--axis', axis, '--steps', '4', '--warmup', '2', '--prewarm_steps', '4', '--repeats', '1', '--no_cpu_baseline',
           '--no_gpu_state', '--min_seconds', '0.05', '--halo_timing_steps', '4']
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=1500)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0, out[-4000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == 8 and d['scaling'] == 'strong' and c['world_size'] == 8 and c['rccl_ranks'] == 0
    assert c['halo_transport'].startswith('peer:' if transport == 'peer' else 'torch.distributed'), c['halo_transport']
    assert '1024x512x512' in c['workload'] and ('128x512x512' if axis == 'x' else '1024x512x64') in c['workload']
    assert sorted(r['rank'] for r in c['per_rank']) == list(range(8))
    assert c['validated'] is True, c['validation']
    assert set(c['validation']) == {'AA', 'AB'}
    for v in c['validation'].values():
        assert v['populations_bit_identical'] and v['ranks_checked'] == 8 and v['rho_rel_err'] < RTOL
        u = v['undivided_box']
        assert u['box'] == '1024x512x512' and u['slabs'] == 8 and u['populations_bit_identical']
        assert u['populations_compared'] == 19 * 2 * 1024 * 512
    print('config 4, 8 ranks on one GPU, %s-slabs, halo transport %s: %.0f MLUPS' % (axis, transport, d['value']))


def _plane(backend, sim, q, z):
    d = sim.desc
    out = np.empty(d.arr_ny * d.arr_nx, dtype=sim.dtype)
    isz = sim.dtype().itemsize
    backend.from_buf(sim.gpu_dist[sim.current_dist_index()] + (q * sim.stride + z * d.arr_ny * d.arr_nx) * isz, out)
    return out.reshape(d.arr_ny, d.arr_nx)


def test_config5_shan_chen_256_full_field():
    """BASELINE config 5 through the host stack (LBSimulationController -> NNSubdomainRunner), 2 steps, every
    population of both lattices and rho / phi / u against the oracle twin."""
    from tests import _host, _sc
    from tests._oracle_group import OracleSCSubdomain
    from sailfish_amd.controller import LBSimulationController
    size, steps = (256, 256, 256), 2
    kw = dict(pattern='AA', fused=True)
    sim_cls, geom = _sc.make_sim(3)
    cfg = _sc.config(3, size, **kw)
    cfg.update(max_iters=steps, quiet=True, perf_stats_every=0)
    ctrl = LBSimulationController(sim_cls, geom, default_config=cfg)
    ctrl.run(ignore_cmdline=True)
    r = ctrl.runners[0]
    sim_cls, geom = _sc.make_sim(3)
    _, _, runners = _host.build_runners(sim_cls, 3, geom, _sc.config(3, size, **kw))
    o = OracleSCSubdomain(runners[0])
    o.run(steps)
    for g_field, o_field in ((r._sim.rho, o.real(o.rho)), (r._sim.phi, o.real(o.phi))):
        assert np.max(np.abs(g_field - o_field) / np.abs(o_field)) < RTOL
    for d in range(3):
        assert np.max(np.abs(r._sim.v[d] - o.real(o.v[d]))) < RTOL * 0.01 + 1e-9
    for grid_num, od in enumerate(o.current()):
        gd = r._debug_get_dist(grid_num=grid_num)
        gd = gd[(slice(None),) + tuple(r._spec._nonghost_slice)]
        assert np.array_equal(gd, o.real(od)), 'lattice %d populations differ' % grid_num
    r.release()
