"""Binary Shan-Chen model on the GPU (config 5 of BASELINE.json, examples/binary_fluid/sc_separation_3d.py)
against the oracle twin: rho, phi, u within 1e-6 relative; populations bit-identical for the linear
pseudopotential."""
import numpy as np
import pytest

from sailfish_amd import sym
from tests import _host, _sc
from tests._oracle_group import OracleSCSubdomain

pytestmark = pytest.mark.gpu


def run_gpu(dim, size, steps, **kw):
    from sailfish_amd.controller import LBSimulationController
    sim_cls, geo = _sc.make_sim(dim)
    cfg = _sc.config(dim, size, **kw)
    cfg.update(max_iters=steps, quiet=True, perf_stats_every=0)
    ctrl = LBSimulationController(sim_cls, geo, default_config=cfg)
    ctrl.run(ignore_cmdline=True)
    return ctrl.runners[0]


def run_oracle(dim, size, steps, **kw):
    sim_cls, geo = _sc.make_sim(dim)
    cfg_, specs, runners = _host.build_runners(sim_cls, dim, geo, _sc.config(dim, size, **kw))
    s = OracleSCSubdomain(runners[0])
    s.run(steps)
    return s


@pytest.mark.parametrize('dim,size', [(2, (70, 20)), (3, (70, 9, 8)), (3, (70, 64, 4))])     # 64 rows: regrouped for the XCDs
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('fused', [True, False])
def test_sc_vs_oracle(dim, size, pattern, fused):
    steps = 21
    r = run_gpu(dim, size, steps, pattern=pattern, fused=fused)
    o = run_oracle(dim, size, steps, pattern=pattern, fused=fused)
    for g_field, o_field in ((r._sim.rho, o.real(o.rho)), (r._sim.phi, o.real(o.phi))):
        assert np.max(np.abs(g_field - o_field) / np.abs(o_field)) < 1e-6
    for d in range(dim):
        assert np.max(np.abs(r._sim.v[d] - o.real(o.v[d]))) < 1e-6 * 0.01 + 1e-9
    for grid_num, od in enumerate(o.current()):
        gd = r._debug_get_dist(grid_num=grid_num)
        gd = gd[(slice(None),) + tuple(r._spec._nonghost_slice)]
        assert np.array_equal(gd, o.real(od)), 'lattice %d populations differ' % grid_num


@pytest.mark.parametrize('dim,size', [(2, (70, 20)), (3, (70, 9, 8)), (3, (130, 16, 4))])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('fused', [True, False])
def test_sc_double_precision(dim, size, pattern, fused):
    """--precision=double through the same kernels (the fused sweep with its own moments keeps both lattices in registers
    there, the aligned pull moves 8-byte values): populations bit-identical to the f64 oracle, fields to 1e-13."""
    steps = 13
    r = run_gpu(dim, size, steps, pattern=pattern, fused=fused, precision='double', G11=-0.2, G22=-0.1)
    o = run_oracle(dim, size, steps, pattern=pattern, fused=fused, precision='double', G11=-0.2, G22=-0.1)
    assert r._sim.rho.dtype == np.float64
    for g_field, o_field in ((r._sim.rho, o.real(o.rho)), (r._sim.phi, o.real(o.phi))):
        assert np.max(np.abs(g_field - o_field) / np.abs(o_field)) < 1e-13
    for d in range(dim):
        assert np.max(np.abs(r._sim.v[d] - o.real(o.v[d]))) < 1e-15
    for grid_num, od in enumerate(o.current()):
        gd = r._debug_get_dist(grid_num=grid_num)
        gd = gd[(slice(None),) + tuple(r._spec._nonghost_slice)]
        assert np.array_equal(gd, o.real(od)), 'lattice %d populations differ' % grid_num


@pytest.mark.parametrize('dim,size', [(2, (70, 20)), (3, (130, 9, 8))])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
def test_fused_sweep_equals_the_two_kernels(dim, size, pattern, monkeypatch):
    """"ShanChenCollideAndPropagateFused" (both lattices in one pass, the stencil sums formed once) and the pair
    "ShanChenPrepareDensities" + "ShanChenCollideAndPropagateFusedV" (the sweep forms its node's densities and velocity
    itself; the default) against the reference's ShanChenPrepareMacroFields + ShanChenCollideAndPropagate0 / 1:
    bit-identical populations and fields, with self-coupling, the classic (exponential) potential and a body force on one
    lattice."""
    from sailfish_amd.controller import LBSimulationController
    out = []
    for mode, fused in (('2', True), ('1', True), ('2', False)):
        monkeypatch.setenv('SLF_SC_FUSED', mode)
        sim_cls, geo = _sc.make_forced_sim(dim, [1e-5, 0.0, 0.0][:dim], [0.0, -2e-5, 0.0][:dim])
        cfg = _sc.config(dim, size, pattern=pattern, fused=True, G12=0.9, G11=-0.3, G22=-0.2, potential='classic',
                         tau_phi=0.8)
        cfg.update(max_iters=11, quiet=True, perf_stats_every=0, hip_sc_fused=fused)
        ctrl = LBSimulationController(sim_cls, geo, default_config=cfg)
        ctrl.run(ignore_cmdline=True)
        r = ctrl.runners[0]
        names = sorted(k.name for pair in r._kernels_none for k in [pair[0]] + list(pair[1]))
        assert ('ShanChenCollideAndPropagateFusedV' in names) == (mode == '2' and fused), names
        assert ('ShanChenPrepareDensities' in names) == (mode == '2' and fused), names
        out.append([r._debug_get_dist(grid_num=g) for g in (0, 1)] + [r._sim.rho.copy(), r._sim.phi.copy()] +
                   [c.copy() for c in r._sim.v])
    for a, b, c in zip(*out):
        assert np.array_equal(a, b, equal_nan=True) and np.array_equal(a, c, equal_nan=True)


def test_densities_pass_stores_the_velocity_on_output_steps_only():
    """The C-ABI contract of "ShanChenPrepareDensities": rho and phi on every launch, vx / vy / vz only when bit 0 of
    `options` is set (the full_output kernels); the fused sweep that follows does not read them."""
    from sailfish_amd.controller import LBSimulationController
    sim_cls, geo = _sc.make_sim(3)
    cfg = _sc.config(3, (70, 9, 8), pattern='AA')
    cfg.update(max_iters=6, quiet=True, perf_stats_every=0)
    ctrl = LBSimulationController(sim_cls, geo, default_config=cfg)
    ctrl.run(ignore_cmdline=True)
    r = ctrl.runners[0]
    b = r.backend
    v_after_run = [c.copy() for c in r._sim.v]
    rho_after_run = r._sim.rho.copy()
    assert any(np.abs(c).max() > 0 for c in v_after_run)
    gpu_v = r.gpu_field(r._sim.v)
    marker = np.float32(7.25)
    for plain in (True, False):
        for c, addr in zip(r._sim.v, gpu_v):
            c[:] = marker
            b.to_buf(addr)
        r._sim.rho[:] = marker
        b.to_buf(r.gpu_field(r._sim.rho))
        it = r._sim.iteration
        b.set_iteration(it)
        kernels = r._kernels_none if plain else r._kernels_full
        b.run_kernel(kernels[it & 1][0], None, r._calc_stream)
        b.sync_stream(r._calc_stream)
        for addr in gpu_v + [r.gpu_field(r._sim.rho)]:
            b.from_buf(addr)
        assert not np.any(r._sim.rho == marker)                                   # the densities: always
        stored = [not np.any(c == marker) for c in r._sim.v]
        assert stored == [not plain] * 3, (plain, stored)


def test_sc_classic_potential_and_self_coupling():
    """exp() differs in the last bits between libm and the GPU: tolerance instead of bit equality."""
    kw = dict(pattern='AA', fused=True, G12=0.9, G11=-0.3, G22=-0.2, potential='classic', tau_phi=0.8)
    r = run_gpu(3, (40, 8, 6), 15, **kw)
    o = run_oracle(3, (40, 8, 6), 15, **kw)
    assert np.max(np.abs(r._sim.rho - o.real(o.rho))) < 1e-6
    assert np.max(np.abs(r._sim.phi - o.real(o.phi))) < 1e-6


def test_sc_phase_separation_and_mass():
    """Physics: the mixture de-mixes at G12 = 1.2 (reference example set-up), each component keeps its mass."""
    r = run_gpu(2, (64, 64), 1500, pattern='AA')
    rho, phi = r._sim.rho.astype(np.float64), r._sim.phi.astype(np.float64)
    assert abs(rho.mean() - 1.0005) < 2e-4 and abs(phi.mean() - 1.0005) < 2e-4
    assert np.abs(rho - phi).max() > 0.5          # separated domains
    assert np.isfinite(rho).all() and np.isfinite(phi).all()


def run_gpu_single(dim, size, steps, **kw):
    from sailfish_amd.controller import LBSimulationController
    sim_cls, geo = _sc.make_single_sim(dim)
    cfg = _sc.single_config(dim, size, **kw)
    cfg.update(max_iters=steps, quiet=True, perf_stats_every=0)
    ctrl = LBSimulationController(sim_cls, geo, default_config=cfg)
    ctrl.run(ignore_cmdline=True)
    return ctrl.runners[0]


@pytest.mark.parametrize('dim,size', [(2, (70, 20)), (3, (70, 9, 8))])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('fused', [True, False])
def test_single_component_vs_oracle(dim, size, pattern, fused):
    from tests._oracle_group import OracleSCSingle
    steps = 21
    kw = dict(pattern=pattern, fused=fused, potential='linear', G=-1.2)
    r = run_gpu_single(dim, size, steps, **kw)
    sim_cls, geo = _sc.make_single_sim(dim)
    cfg_, specs, runners = _host.build_runners(sim_cls, dim, geo, _sc.single_config(dim, size, **kw))
    o = OracleSCSingle(runners[0])
    o.run(steps)
    assert np.array_equal(r._sim.rho, o.real(o.rho))
    for d in range(dim):
        assert np.array_equal(r._sim.v[d], o.real(o.v[d]))
    gd = r._debug_get_dist()[(slice(None),) + tuple(r._spec._nonghost_slice)]
    assert np.array_equal(gd, o.real(o.current()))


@pytest.mark.parametrize('G', [3.5, 4.5, 5.0, 5.5])
def test_phase_separation_matches_reference_record(G, golden_dir):
    """The reference's recorded regression data for this very set-up (regtest/sc_phase_sep.py ->
    regtest/results/sc_phase_separation/single.dat: coupling G, min rho, max rho, order parameter after
    50000 steps of examples/sc_phase_separation.py, seed 2348).  Below the spinodal point (|G| < ~4) the
    fluid stays homogeneous, above it separates into coexisting liquid / vapour densities."""
    import os
    data = np.loadtxt(os.path.join(golden_dir, 'sc_phase_separation_single.dat'))
    row = data[np.argmin(np.abs(data[:, 0] - G))]
    assert abs(row[0] - G) < 1e-6
    r = run_gpu_single(2, (256, 256), 50000, pattern='AB', G=-G, potential='classic')
    rho = r._sim.rho
    lo, hi = float(rho.min()), float(rho.max())
    if row[2] - row[1] < 1e-3:           # homogeneous state
        assert hi - lo < 1e-3 and abs(0.5 * (lo + hi) - 0.5 * (row[1] + row[2])) < 2e-3
    else:                                 # coexistence densities (droplet curvature / coarsening stage: few %)
        assert abs(hi - row[2]) / row[2] < 0.03, (lo, hi, row)
        assert abs(lo - row[1]) < 0.02, (lo, hi, row)


@pytest.mark.parametrize('single', [False, True])
@pytest.mark.parametrize('dim,size,nsub,axis', [(2, (70, 20), 2, 'x'), (3, (40, 9, 8), 2, 'z'), (3, (40, 12, 8), 3, 'y')])
@pytest.mark.parametrize('pattern,fused', [('AB', True), ('AA', True), ('AA', False)])
def test_sc_multi_subdomain(single, dim, size, nsub, axis, pattern, fused):
    """Non-local models on several subdomains (macro-field halo + population halo of every lattice;
    reference NNSubdomainRunner, regtest/subdomains/binary_pbc.py): equal to the oracle group, which the
    CPU suite shows to be bit-identical to a single subdomain."""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    from tests._oracle_group import OracleNNGroup
    steps = 11
    sim_cls, _ = (_sc.make_single_sim if single else _sc.make_sim)(dim)
    cfg = (_sc.single_config if single else _sc.config)(dim, size, pattern=pattern, fused=fused)
    if single:
        cfg.update(G=-1.2, sc_potential='linear')
    cfg.update(subdomains=nsub, conn_axis=axis)
    geo_name = 'EqualSubdomainsGeometry%dD' % dim
    og = OracleNNGroup(sim_cls, dim, geo_name, dict(cfg), single=single)
    og.run(steps)
    gcfg = dict(cfg, max_iters=steps, quiet=True, perf_stats_every=0)
    ctrl = LBSimulationController(sim_cls, getattr(geo_mod, geo_name), default_config=gcfg)
    ctrl.run(ignore_cmdline=True)
    assert len(ctrl.runners) == nsub
    for r, o in zip(ctrl.runners, og.subs):
        assert r._spec.id == o.runner._spec.id
        assert np.array_equal(r._sim.rho, o.real(o.rho))
        lattices = [o.current()] if single else list(o.current())
        for grid_num, od in enumerate(lattices):
            gd = r._debug_get_dist(grid_num=grid_num)[(slice(None),) + tuple(r._spec._nonghost_slice)]
            assert np.array_equal(gd, o.real(od)), 'subdomain %d lattice %d' % (r._spec.id, grid_num)


@pytest.mark.parametrize('mode', ['one stream', 'events', 'copies'])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('size,nsub,precision,potential', [((40, 9, 8), 2, 'single', 'linear'), ((48, 12, 8), 3, 'single', 'classic'),
                                                           ((130, 16, 6), 2, 'double', 'linear'), ((264, 34, 5), 4, 'single', 'linear')])
def test_sc_x_slabs_through_planes(size, nsub, precision, potential, pattern, mode, monkeypatch):
    """1-D decomposition along x of the binary model: the populations of both lattices and the densities cross the faces
    through the dense planes the two kernels of a step write and read themselves (slf_module_set_xface_planes,
    xface.NNPlanes) -- no ghost columns, no pack / unpack launches.  Equal to the oracle group bit for bit (populations of
    both lattices after materialise(), rho, phi); the subdomains of one process share the planes and run on one stream
    ('one stream'), order them with events (SLF_GROUP_ONE_STREAM=0) or copy them (SLF_XFACE_SHARE=0).  Both access
    patterns; 11 steps: the in-place run ends after an even step, whose crossings belong into the ghost columns."""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    from tests._oracle_group import OracleNNGroup
    if mode == 'events':
        monkeypatch.setenv('SLF_GROUP_ONE_STREAM', '0')
    elif mode == 'copies':
        monkeypatch.setenv('SLF_XFACE_SHARE', '0')
    steps = 12 if nsub == 3 else 11          # in place: once ending after an odd (push) step, else after an even one
    sim_cls, _ = _sc.make_sim(3)
    cfg = _sc.config(3, size, pattern=pattern, fused=True, precision=precision, potential=potential)
    cfg.update(subdomains=nsub, conn_axis='x')
    og = OracleNNGroup(sim_cls, 3, 'EqualSubdomainsGeometry3D', dict(cfg))
    og.run(steps)
    gcfg = dict(cfg, max_iters=steps, quiet=True, perf_stats_every=0)
    ctrl = LBSimulationController(sim_cls, geo_mod.EqualSubdomainsGeometry3D, default_config=gcfg)
    ctrl.run(ignore_cmdline=True)
    assert len(ctrl.runners) == nsub
    tol = dict(rtol=0, atol=0) if potential == 'linear' else dict(rtol=2e-6, atol=1e-7)      # expf: the device's own
    for r, o in zip(ctrl.runners, og.subs):
        assert r._nnx is not None and r._nnx.shared == (mode != 'copies')
        assert not r._links[sorted(r._links)[0]].kernels[('push', 0)][0]        # nothing is packed
        if potential == 'linear':
            assert np.array_equal(r._sim.rho, o.real(o.rho)) and np.array_equal(r._sim.phi, o.real(o.phi))
        else:
            np.testing.assert_allclose(r._sim.rho, o.real(o.rho), **tol)
        for grid_num, od in enumerate(o.current()):
            gd = r._debug_get_dist(grid_num=grid_num)[(slice(None),) + tuple(r._spec._nonghost_slice)]
            if potential == 'linear':
                assert np.array_equal(gd, o.real(od)), 'subdomain %d lattice %d' % (r._spec.id, grid_num)
            else:
                np.testing.assert_allclose(gd, o.real(od), **tol)


@pytest.mark.parametrize('mode', ['one stream', 'events', 'copies'])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('size,nsub,precision', [((40, 9, 8), 2, 'single'), ((200, 34, 5), 3, 'single'), ((130, 16, 6), 2, 'double')])
def test_single_component_x_slabs_through_planes(size, nsub, precision, pattern, mode, monkeypatch):
    """The single-component model (PrepareMacroFields + CollideAndPropagate, reference lb_single.py:242-347) over the same
    planes: one lattice, field 0 of the density planes.  Equal to the oracle group bit for bit, both access patterns, an
    odd and an even number of steps, the three orderings of a same-process group."""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    from tests._oracle_group import OracleNNGroup
    if mode == 'events':
        monkeypatch.setenv('SLF_GROUP_ONE_STREAM', '0')
    elif mode == 'copies':
        monkeypatch.setenv('SLF_XFACE_SHARE', '0')
    steps = 12 if nsub == 3 else 11
    sim_cls, _ = _sc.make_single_sim(3)
    cfg = _sc.single_config(3, size, pattern=pattern, fused=True)
    cfg.update(G=-1.2, sc_potential='linear', precision=precision, subdomains=nsub, conn_axis='x')
    og = OracleNNGroup(sim_cls, 3, 'EqualSubdomainsGeometry3D', dict(cfg), single=True)
    og.run(steps)
    gcfg = dict(cfg, max_iters=steps, quiet=True, perf_stats_every=0)
    ctrl = LBSimulationController(sim_cls, geo_mod.EqualSubdomainsGeometry3D, default_config=gcfg)
    ctrl.run(ignore_cmdline=True)
    assert len(ctrl.runners) == nsub
    for r, o in zip(ctrl.runners, og.subs):
        assert r._nnx is not None and r._nnx.n_lat == 1 and r._nnx.shared == (mode != 'copies')
        assert np.array_equal(r._sim.rho, o.real(o.rho))
        gd = r._debug_get_dist(grid_num=0)[(slice(None),) + tuple(r._spec._nonghost_slice)]
        assert np.array_equal(gd, o.real(o.current())), 'subdomain %d' % r._spec.id


@pytest.mark.parametrize('mode', ['one stream', 'events', 'copies'])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('single', [False, True])
def test_sc_walls_x_slabs_through_planes(single, pattern, mode, monkeypatch):
    """Planes with a node map: a mixture (a vapour) between two solid slabs, y not periodic, a solid block that straddles
    the seam between the first two of three x-slabs.  Dry nodes never store a density into the planes (those entries are
    primed once from the fields, as a ghost column would have received them once), excluded edge nodes send nothing
    (markers), bounce-back nodes on the seam swap what came through the planes -- every wet node equal to the oracle
    group bit for bit."""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    from tests._oracle_group import OracleNNGroup
    if mode == 'events':
        monkeypatch.setenv('SLF_GROUP_ONE_STREAM', '0')
    elif mode == 'copies':
        monkeypatch.setenv('SLF_XFACE_SHARE', '0')
    steps = 12 if pattern == 'AA' else 11
    size = (24, 22, 8)
    if single:
        sim_cls, _ = _sc.make_single_wall_sim(3)
        cfg = _sc.single_config(3, size, pattern=pattern)
        cfg.update(G=-1.2, sc_potential='linear')
    else:
        sim_cls, _ = _sc.make_wall_sim(3)
        cfg = _sc.config(3, size, pattern=pattern)
    cfg.update(periodic_y=False, subdomains=3, conn_axis='x')
    og = OracleNNGroup(sim_cls, 3, 'EqualSubdomainsGeometry3D', dict(cfg), single=single)
    og.run(steps)
    ctrl = LBSimulationController(sim_cls, geo_mod.EqualSubdomainsGeometry3D,
                                  default_config=dict(cfg, max_iters=steps, quiet=True, perf_stats_every=0))
    ctrl.run(ignore_cmdline=True)
    assert len(ctrl.runners) == 3
    for r, o in zip(ctrl.runners, og.subs):
        assert r._nnx is not None and not r._desc.fluid_only
        wet = r._subdomain.fluid_map()
        assert wet.any() and not wet.all()
        assert np.array_equal(r._sim.rho[wet], o.real(o.rho)[wet])
        lattices = [o.current()] if single else list(o.current())
        for grid_num, od in enumerate(lattices):
            gd = r._debug_get_dist(grid_num=grid_num)[(slice(None),) + tuple(r._spec._nonghost_slice)]
            assert np.array_equal(gd[:, wet], o.real(o.dense(od))[:, wet]), 'subdomain %d lattice %d' % (r._spec.id, grid_num)
    # the block does sit on the seam: both sides of it hold dry nodes in their edge columns
    assert not ctrl.runners[0]._subdomain.fluid_map()[..., -1].all() and not ctrl.runners[1]._subdomain.fluid_map()[..., 0].all()


@pytest.mark.parametrize('single,pattern', [(False, 'AB'), (False, 'AA'), (True, 'AB')])
def test_sc_closed_box_x_slabs_through_planes(single, pattern):
    """x NOT periodic: the first and the last of three x-slabs have one connected face each (planes) and one wall (ghost
    columns, never read by a node that computes a force); walls along y too, z wrapped in-kernel.  Every wet node equal to
    the oracle group bit for bit.  (Not the single-component model in place: its density pass also visits the wall nodes
    -- the reference's PrepareMacroFields skips excluded nodes only, lb_single_fluid.mako:143-146 -- and in the odd step a
    wall node on the rim of the lattice pulls from the ghost layer, whose populations are the equilibrium of the +inf every
    field holds outside the lattice: an infinite wall density, NaN forces next to it.  With or without planes, on one
    subdomain or three: a property of that combination)"""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    from sailfish_amd.node_type import NTFullBBWall
    from tests._oracle_group import OracleNNGroup
    base, _ = (_sc.make_single_sim if single else _sc.make_sim)(3)

    class Boxed(base.subdomain):
        def boundary_conditions(self, hx, hy, hz):
            wall = (hx == 0) | (hx == self.gx - 1) | (hy == 0) | (hy == self.gy - 1)
            self.set_node(wall, NTFullBBWall)

    class Sim(base):
        subdomain = Boxed

    size = (30, 12, 8)
    cfg = (_sc.single_config if single else _sc.config)(3, size, pattern=pattern)
    if single:
        cfg.update(G=-1.2, sc_potential='linear')
    cfg.update(periodic_x=False, periodic_y=False, subdomains=3, conn_axis='x')
    steps = 12 if pattern == 'AA' else 11
    og = OracleNNGroup(Sim, 3, 'EqualSubdomainsGeometry3D', dict(cfg), single=single)
    og.run(steps)
    ctrl = LBSimulationController(Sim, geo_mod.EqualSubdomainsGeometry3D,
                                  default_config=dict(cfg, max_iters=steps, quiet=True, perf_stats_every=0))
    ctrl.run(ignore_cmdline=True)
    assert len(ctrl.runners) == 3
    from sailfish_amd import xface
    for i, (r, o) in enumerate(zip(ctrl.runners, og.subs)):
        assert r._nnx is not None
        connected = [bool(r._nnx.recv['macro'][0][f]) for f in (xface.LOW, xface.HIGH)]
        assert connected == [i > 0, i < 2]
        wet = r._subdomain.fluid_map()
        assert wet.any() and not wet.all()
        assert np.array_equal(r._sim.rho[wet], o.real(o.rho)[wet])
        lattices = [o.current()] if single else list(o.current())
        for grid_num, od in enumerate(lattices):
            gd = r._debug_get_dist(grid_num=grid_num)[(slice(None),) + tuple(r._spec._nonghost_slice)]
            assert np.array_equal(gd[:, wet], o.real(o.dense(od))[:, wet]), 'subdomain %d lattice %d' % (r._spec.id, grid_num)


def test_sc_x_slab_planes_refuse_what_they_do_not_serve():
    """The planes are for the kernels that know them: a module with indirect addressing does not take them, and a runner
    whose y / z periodicity is made by the ghost-layer kernels (images that live in the arrays, not in the planes) keeps
    the ghost columns."""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    sim_cls, _ = _sc.make_sim(3)
    for extra, library_refuses in ((dict(node_addressing='indirect'), True), (dict(hip_fused_periodic=False), False)):
        cfg = _sc.config(3, (40, 9, 8))
        cfg.update(subdomains=2, conn_axis='x', max_iters=3, quiet=True, perf_stats_every=0)
        cfg.update(extra)
        ctrl = LBSimulationController(sim_cls, geo_mod.EqualSubdomainsGeometry3D, default_config=cfg)
        ctrl.run(ignore_cmdline=True)
        for r in ctrl.runners:
            assert r._nnx is None
            if library_refuses:
                with pytest.raises(Exception):
                    r.backend.set_xface_planes(r.module, 2, 1 << 20, 0, 1 << 21, 0)


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('steps_before', [8, 7])
def test_sc_x_slab_planes_checkpoint_roundtrip(tmp_path, steps_before, pattern):
    """Three x-slabs of the binary model exchanging through planes: a checkpoint holds what has CROSSED the faces too (the
    arrays are stale there until NNPlanes.materialise() writes the received planes into them), and a restored run starts
    from the arrays alone (planes reset to 'nothing crossed') -- steps + restore + the rest == a straight run of 15, every
    population of both lattices of every subdomain bit for bit; after an even and after an odd number of steps (the two
    copies of the two-copy pattern, the two parities of the planes; in place: a restart before an even step, which reads
    the first columns, and before an odd one, which pulls out of the ghost columns)."""
    import os
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    sim_cls, _ = _sc.make_sim(3)

    def run(steps, **extra):
        cfg = _sc.config(3, (48, 10, 8), pattern=pattern)
        cfg.update(max_iters=steps, quiet=True, perf_stats_every=0, subdomains=3, conn_axis='x', **extra)
        ctrl = LBSimulationController(sim_cls, geo_mod.EqualSubdomainsGeometry3D, default_config=cfg)
        ctrl.run(ignore_cmdline=True)
        assert all(r._nnx is not None for r in ctrl.runners)
        return ctrl.runners

    ck = str(tmp_path / 'ck')
    run(steps_before, checkpoint_file=ck, final_checkpoint=True)
    files = sorted(f for f in os.listdir(str(tmp_path)) if f.endswith('.cpoint.npz'))
    assert len(files) == 3
    cont = run(15, restore_from=ck + '.last')
    ref = run(15)
    for a, b in zip(cont, ref):
        assert a._sim.iteration == 15
        for g in (0, 1):
            assert np.array_equal(a._debug_get_dist(grid_num=g), b._debug_get_dist(grid_num=g), equal_nan=True)
        assert np.array_equal(a._sim.rho, b._sim.rho) and np.array_equal(a._sim.phi, b._sim.phi)


@pytest.mark.parametrize('dim,size', [(2, (70, 20)), (3, (40, 9, 8))])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
def test_sc_per_lattice_body_force(dim, size, pattern):
    """Body forces acting on one lattice each (reference add_body_force(..., grid=k),
    examples/binary_fluid/sc_poiseuille_2d.py): GPU == oracle, populations bit-identical."""
    from sailfish_amd.controller import LBSimulationController
    a0, a1 = [2e-5, 0.0, -1e-5][:dim], [0.0, 3e-5, 1e-5][:dim]
    sim_cls, geo = _sc.make_forced_sim(dim, a0, a1)
    cfg = _sc.config(dim, size, pattern=pattern)
    ocfg_, specs, runners = _host.build_runners(sim_cls, dim, geo, dict(cfg))
    o = OracleSCSubdomain(runners[0])
    o.run(15)
    gcfg = dict(cfg, max_iters=15, quiet=True, perf_stats_every=0)
    ctrl = LBSimulationController(sim_cls, geo, default_config=gcfg)
    ctrl.run(ignore_cmdline=True)
    r = ctrl.runners[0]
    assert list(r._desc.accel1)[:dim] == a1
    for grid_num, od in enumerate(o.current()):
        gd = r._debug_get_dist(grid_num=grid_num)[(slice(None),) + tuple(r._spec._nonghost_slice)]
        assert np.array_equal(gd, o.real(od)), 'lattice %d' % grid_num


@pytest.mark.parametrize('dim,size', [(2, (70, 20)), (3, (40, 9, 8))])
def test_sc_edm(dim, size):
    """Shan-Chen force through the exact difference method (the setting of the reference's
    examples/binary_fluid/sc_capillary_wave_2d.py, sc_laplace_2d.py, sc_poiseuille_2d.py)."""
    from sailfish_amd.controller import LBSimulationController
    sim_cls, geo = _sc.make_forced_sim(dim, None, [0.0, 2e-5, 0.0][:dim])
    cfg = _sc.config(dim, size, pattern='AA')
    cfg['force_implementation'] = 'edm'
    ocfg_, specs, runners = _host.build_runners(sim_cls, dim, geo, dict(cfg))
    o = OracleSCSubdomain(runners[0])
    assert o.desc.force_implementation == 1
    o.run(15)
    ctrl = LBSimulationController(sim_cls, geo, default_config=dict(cfg, max_iters=15, quiet=True, perf_stats_every=0))
    ctrl.run(ignore_cmdline=True)
    r = ctrl.runners[0]
    for grid_num, od in enumerate(o.current()):
        gd = r._debug_get_dist(grid_num=grid_num)[(slice(None),) + tuple(r._spec._nonghost_slice)]
        assert np.array_equal(gd, o.real(od)), 'lattice %d' % grid_num
    # and it is a different scheme from Guo's
    cfg['force_implementation'] = 'guo'
    ocfg_, specs, runners = _host.build_runners(sim_cls, dim, geo, dict(cfg))
    g = OracleSCSubdomain(runners[0])
    g.run(15)
    assert not np.array_equal(g.real(g.current()[0]), o.real(o.current()[0]))


@pytest.mark.parametrize('nx', [40, 64, 128, 200, 256, 512, 1024, 1100])    # 1100: three segments (slf_rowpush.h: row_block_x); 40, 200: a partial last wave
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('single', [False, True])
def test_sc_full_wave_rows(nx, pattern, single):
    """Shan-Chen row sweeps (slf_sc.hip on top of row_push) at rows that fill their wavefronts exactly
    (nx = 64 k): x == nx is lane 63 of the last wave and the seam exchange has no idle lane to hide behind.
    Bit-identical to the oracle (linear pseudopotential)."""
    from tests._oracle_group import OracleSCSingle
    steps, size = 7, (nx, 4, 3)
    if single:
        kw = dict(pattern=pattern, fused=True, potential='linear', G=-1.2)
        r = run_gpu_single(3, size, steps, **kw)
        sim_cls, geo = _sc.make_single_sim(3)
        cfg_, specs, runners = _host.build_runners(sim_cls, 3, geo, _sc.single_config(3, size, **kw))
        o = OracleSCSingle(runners[0])
        o.run(steps)
        assert np.array_equal(r._sim.rho, o.real(o.rho))
        gd = r._debug_get_dist()[(slice(None),) + tuple(r._spec._nonghost_slice)]
        assert np.array_equal(gd, o.real(o.current()))
        return
    r = run_gpu(3, size, steps, pattern=pattern, fused=True)
    o = run_oracle(3, size, steps, pattern=pattern, fused=True)
    assert np.array_equal(r._sim.rho, o.real(o.rho)) and np.array_equal(r._sim.phi, o.real(o.phi))
    for grid_num, od in enumerate(o.current()):
        gd = r._debug_get_dist(grid_num=grid_num)
        gd = gd[(slice(None),) + tuple(r._spec._nonghost_slice)]
        assert np.array_equal(gd, o.real(od)), 'lattice %d populations differ' % grid_num


@pytest.mark.parametrize('grid', [sym.D2Q9, sym.D3Q19])
@pytest.mark.parametrize('potential', ['linear', 'classic'])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
def test_sc_kernels_against_reference_composition(grid, potential, pattern, golden_dir):
    """The HIP Shan-Chen kernels against the fixtures composed from the reference's own objects, without the oracle
    in between (the GPU twin of tests/test_sc_oracle.py::test_node_update_matches_reference_composition):
    ShanChenPrepareMacroFields, then ShanChenCollideAndPropagate0 / 1 on a 3^dim block of fluid nodes whose centre
    carries the sample's populations and whose neighbours carry the sample's rho / phi.  f64, <= 1e-13."""
    import os
    from sailfish_amd import hipabi
    from sailfish_amd.backend_hip import HIPBackend

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    g = np.load(os.path.join(golden_dir, 'shan_chen_%s.npz' % grid.__name__))
    dim, Q = grid.dim, grid.Q
    e = grid.basis_array
    visc, tau_phi = float(g['visc'][0]), float(g['tau_phi'][0])
    nz = 5 if dim == 3 else 1
    aa = pattern == 'AA'
    worst = 0.0
    w = np.array(grid.weights_float)
    stream = b.make_stream()
    for k in range(len(g['f1'])):
        desc = hipabi.make_desc(lattice=grid.slf_id, model=hipabi.SLF_BGK, precision=8,
                                access_pattern=hipabi.SLF_AA if aa else hipabi.SLF_AB,
                                lat_nx=5, lat_ny=5, lat_nz=nz, arr_nx=32, arr_ny=5, arr_nz=nz, fluid_only=1,
                                tau=sym.relaxation_time(visc), visc=visc, tau_phi=tau_phi,
                                simtype=hipabi.SLF_SIM_SHAN_CHEN_BINARY, sc_G=list(g['G'][k]),
                                sc_potential=0 if potential == 'linear' else 1,
                                accel=list(g['body_accel'][k, 0]) + [0.0] * (3 - dim),
                                accel1=list(g['body_accel'][k, 1]) + [0.0] * (3 - dim),
                                has_force=int(np.any(g['body_accel'][k] != 0.0)))
        module = b.build(desc)
        shape = (nz, 5, 32)
        nodes, stride = nz * 5 * 32, hipabi.dist_stride(desc)
        c = (2, 2, 2) if dim == 3 else (0, 2, 2)
        host = []
        for f in (g['f1'][k], g['f2'][k]):
            raw = np.zeros((Q, stride))
            d = raw[:, :nodes].reshape((Q,) + shape)
            d[...] = w.reshape((Q, 1, 1, 1))
            d[(slice(None),) + c] = f
            host.append(raw)
        off = b.dist_align_offset(8)
        dev_in = [b.alloc_buf(size=Q * stride * 8, align_offset=off) for _ in range(2)]
        dev_out = dev_in if aa else [b.alloc_buf(size=Q * stride * 8, align_offset=off) for _ in range(2)]
        for addr, raw in zip(dev_in, host):
            b.to_buf(addr, raw)
        if not aa:
            for addr in dev_out:
                b.to_buf(addr, np.zeros((Q, stride)))
        fields = [np.full(shape, 1.0), np.full(shape, 1.0)] + [np.zeros(shape) for _ in range(dim)]
        dev_f = [b.alloc_buf(like=a) for a in fields]
        sig = 'P' * (5 + dim) + 'i'

        def kern(name, a, o):
            return b.get_kernel(module, name, (64,), [0, a, o] + dev_f + [0], sig, needs_iteration=aa)
        b.set_iteration(0)                                   # even AA step / AB: the node's own slots hold f_i
        b.run_kernel(kern('ShanChenPrepareMacroFields', dev_in[0], dev_in[1]), None, stream)
        stream.synchronize()
        for addr in dev_f:
            b.from_buf(addr)
        rho, phi, v = fields[0], fields[1], fields[2:]
        assert abs(rho[c] - g['sc_rho'][k]) < 1e-14 and abs(phi[c] - g['sc_phi'][k]) < 1e-14
        for a in range(dim):
            assert abs(v[a][c] - g['sc_v'][k, a]) < 1e-15
        for i in range(1, Q):                                 # the neighbours' densities as in the sample
            p = (c[0] + (e[i][2] if dim == 3 else 0), c[1] + e[i][1], c[2] + e[i][0])
            rho[p], phi[p] = g['rho_nb'][k, i], g['phi_nb'][k, i]
        b.to_buf(dev_f[0])
        b.to_buf(dev_f[1])
        post = g['sc_post_' + potential][k]
        for l in range(2):
            b.run_kernel(kern('ShanChenCollideAndPropagate%d' % l, dev_in[l], dev_out[l]), None, stream)
            stream.synchronize()
            raw = np.zeros((Q, stride))
            b.from_buf(dev_out[l], raw)
            dout = raw[:, :nodes].reshape((Q,) + shape)
            for i in range(Q):
                if aa:                                        # in place, opposite slot
                    got = dout[(grid.idx_opposite[i],) + c]
                else:                                         # pushed to x + e_i
                    p = (c[0] + (e[i][2] if dim == 3 else 0), c[1] + e[i][1], c[2] + e[i][0])
                    got = dout[(i,) + p]
                worst = max(worst, abs(got - post[l, i]))
        for addr in set(dev_in + dev_out + dev_f):
            b.free_buf(addr)
    assert worst < 1e-13, worst


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('dim,size,nsub,axis', [(2, (70, 26), 1, 'x'), (3, (40, 22, 8), 1, 'x'), (3, (40, 22, 8), 2, 'x'),
                                                (3, (24, 22, 12), 2, 'z')])
def test_sc_indirect_addressing(dim, size, nsub, axis, pattern):
    """--node_addressing=indirect with the binary model (reference lb_binary.py:457-465, subdomain_runner.py:829-878
    serves NNSubdomainRunner too): both lattices hold the active nodes only, rho / phi / u stay dense.  Equal to the
    oracle group run with sparse arrays, which tests/test_indirect_oracle.py shows to equal the dense run."""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    from tests._oracle_group import OracleNNGroup
    steps = 11
    sim_cls, _ = _sc.make_wall_sim(dim)
    cfg = _sc.config(dim, size, pattern=pattern)
    cfg.update(periodic_y=False, subdomains=nsub, conn_axis=axis, node_addressing='indirect')
    geo_name = 'EqualSubdomainsGeometry%dD' % dim
    og = OracleNNGroup(sim_cls, dim, geo_name, dict(cfg))
    og.run(steps)
    ctrl = LBSimulationController(sim_cls, getattr(geo_mod, geo_name),
                                  default_config=dict(cfg, max_iters=steps, quiet=True, perf_stats_every=0))
    ctrl.run(ignore_cmdline=True)
    assert len(ctrl.runners) == nsub
    for r, o in zip(ctrl.runners, og.subs):
        assert r._desc.node_addressing == 1 and r._dist_stride < 0.85 * int(np.prod(r._physical_size))
        assert 0.5 < r._subdomain.active_node_mask.mean() < 0.9
        wet = r._subdomain.fluid_map()
        assert np.array_equal(r._sim.rho[wet], o.real(o.rho)[wet])
        assert np.array_equal(r._sim.phi[wet], o.real(o.phi)[wet])
        for grid_num, od in enumerate(o.current()):
            gd = r._debug_get_dist(grid_num=grid_num)[(slice(None),) + tuple(r._spec._nonghost_slice)]
            assert np.array_equal(gd[:, wet], o.real(o.dense(od))[:, wet]), 'subdomain %d lattice %d' % (r._spec.id, grid_num)


@pytest.mark.parametrize('mode', ['2', '1'])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('dim,size,nsub,axis', [(2, (70, 26), 1, 'x'), (3, (70, 22, 8), 1, 'x'), (3, (140, 22, 8), 1, 'x'),
                                                (3, (40, 22, 8), 2, 'x'), (3, (40, 22, 12), 2, 'z')])
def test_sc_walls_dense_addressing(dim, size, nsub, axis, pattern, mode, monkeypatch):
    """The binary mixture between solid slabs with a solid block in the channel, dense arrays: the node-map instantiations
    of the fused sweeps (rows wrapped along x: the aligned pull of the odd step with bounce-back and dry nodes in the
    row; x-split: the x-shifted loads) against the oracle group, every wet node bit for bit."""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    from tests._oracle_group import OracleNNGroup
    monkeypatch.setenv('SLF_SC_FUSED', mode)
    steps = 11
    sim_cls, _ = _sc.make_wall_sim(dim)
    cfg = _sc.config(dim, size, pattern=pattern)
    cfg.update(periodic_y=False, subdomains=nsub, conn_axis=axis)
    geo_name = 'EqualSubdomainsGeometry%dD' % dim
    og = OracleNNGroup(sim_cls, dim, geo_name, dict(cfg))
    og.run(steps)
    ctrl = LBSimulationController(sim_cls, getattr(geo_mod, geo_name),
                                  default_config=dict(cfg, max_iters=steps, quiet=True, perf_stats_every=0))
    ctrl.run(ignore_cmdline=True)
    assert len(ctrl.runners) == nsub
    for r, o in zip(ctrl.runners, og.subs):
        wet = r._subdomain.fluid_map()
        assert wet.any() and not wet.all()
        assert np.array_equal(r._sim.rho[wet], o.real(o.rho)[wet])
        assert np.array_equal(r._sim.phi[wet], o.real(o.phi)[wet])
        for d in range(dim):
            assert np.max(np.abs(r._sim.v[d][wet] - o.real(o.v[d])[wet])) < 1e-9
        for grid_num, od in enumerate(o.current()):
            gd = r._debug_get_dist(grid_num=grid_num)[(slice(None),) + tuple(r._spec._nonghost_slice)]
            assert np.array_equal(gd[:, wet], o.real(o.dense(od))[:, wet]), 'subdomain %d lattice %d' % (r._spec.id, grid_num)


@pytest.mark.parametrize('pattern,walls', [('AB', True), ('AA', False), ('AB', False)])
@pytest.mark.parametrize('dim,size,nsub', [(2, (70, 26), 1), (3, (40, 22, 8), 1), (3, (40, 22, 8), 2)])
def test_single_component_indirect_addressing(dim, size, nsub, pattern, walls):
    """--node_addressing=indirect with the single-component Shan-Chen model (the reference's sparse address map serves
    every runner, subdomain_runner.py:829-878; PrepareMacroFields and CollideAndPropagate take the `nodes` table as their
    first argument): the distributions hold the active nodes only, the density field stays dense -- and every wet node
    carries the same populations and density as in the dense run, bit for bit.  With solid walls only in the two-copy
    pattern: PrepareMacroFields also computes a density for the WALL nodes next to the fluid (lb_single_fluid.mako:
    143-146 skips excluded nodes only), and in the in-place pattern their odd step pulls from the wall's inner nodes,
    which hold no slot in the sparse arrays -- the reference indexes `nodes[]` unguarded there (geo_helpers.mako:246-252);
    here such a population counts as 0, so the pseudopotential of those wall nodes is a different (equally arbitrary)
    number than in the dense run."""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    steps = 11
    sim_cls, _ = _sc.make_single_wall_sim(dim) if walls else _sc.make_single_sim(dim)
    cfg = _sc.single_config(dim, size, pattern=pattern, G=-1.2, potential='linear')
    cfg.update(periodic_y=not walls, subdomains=nsub, conn_axis='x')
    geo_name = 'EqualSubdomainsGeometry%dD' % dim
    runs = []
    for addressing in ('indirect', 'direct'):
        ctrl = LBSimulationController(sim_cls, getattr(geo_mod, geo_name),
                                      default_config=dict(cfg, node_addressing=addressing, max_iters=steps, quiet=True,
                                                          perf_stats_every=0))
        ctrl.run(ignore_cmdline=True)
        runs.append(ctrl)
    sparse, dense = runs
    assert len(sparse.runners) == nsub
    for rs, rd in zip(sparse.runners, dense.runners):
        assert rs._desc.node_addressing == 1
        if walls:
            assert rs._dist_stride < 0.85 * int(np.prod(rs._physical_size))
        wet = rs._subdomain.fluid_map()
        assert wet.any() and np.array_equal(wet, rd._subdomain.fluid_map())
        assert np.array_equal(rs._sim.rho[wet], rd._sim.rho[wet])
        sl = (slice(None),) + tuple(rs._spec._nonghost_slice)
        assert np.array_equal(rs._debug_get_dist()[sl][:, wet], rd._debug_get_dist()[sl][:, wet])
