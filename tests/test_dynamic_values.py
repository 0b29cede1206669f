"""Boundary values that depend on position and time (reference node_type.py:471-626 DynamicValue,
LinearlyInterpolatedTimeSeries; boundary.mako:52-84), evaluated on the HOST here: the classes, the encoder's table
entries, and the per-step updates -- on the CPU with the oracle twin of the runner (tests/_oracle_group.py)."""
import numpy as np
import pytest

from sailfish_amd import node_type as nt
from sailfish_amd import sym
from sailfish_amd.geo import LBGeometry2D
from tests import _host


def test_time_series_semantics_of_the_reference():
    """reference tests/node_type.py:6-30 (equality and hashing of series) and boundary.mako:52-76 (the interpolation:
    wrapped, linear, `step_size` iterations between two data points)."""
    L = nt.LinearlyInterpolatedTimeSeries
    a, b, c = L([1, 2, 3], 2.0), L((1, 2), 3.0), L(np.float32([4, 5, 6]))
    assert a != b and a != c and b != c
    a, b, c = L([1, 2, 3], 2.0), L([1, 2, 3], 3.0), L(np.float64([1, 2, 3]), 2.0)
    assert a == c and a != b and hash(a) == hash(c) and len({a, b, c}) == 2
    data = [1.0, 4.0, 2.0]
    s = L(data, 5.0)
    for it, want in ((0, 1.0), (5, 4.0), (10, 2.0), (15, 1.0), (2, 2.2), (12, 1.6), (14, 1.2), (29, 1.2), (31, 1.6)):
        assert abs(s.at(it) - want) < 1e-12, (it, s.at(it))
    assert nt.timeseries_interpolate(data, 5.0, 7) == 0.4 * 2.0 + 4.0 * (1 - 0.4)


def test_dynamic_value_evaluation():
    import sympy
    S = sym.S
    d = nt.DynamicValue(0.1 * S.gx * nt.LinearlyInterpolatedTimeSeries([0.0, 1.0], 40), 0.0)
    assert d.time_dependent() and d.space_dependent() and len(d) == 2 and d.has_symbols(S.gx) and not d.has_symbols(S.gy)
    v = d.evaluate((np.array([0.0, 2.0, 5.0]), np.zeros(3)), 10, 1.0)
    assert np.allclose(v, [[0, 0], [0.05, 0], [0.125, 0]])
    p = nt.DynamicValue(1.0 + 0.01 * sympy.sin(S.time * 0.25))
    assert p.time_dependent() and not p.space_dependent()
    assert np.allclose(p.evaluate((np.zeros(1), np.zeros(1)), 8, 0.5), 1.0 + 0.01 * np.sin(1.0))     # time = iteration x dt
    q = nt.DynamicValue(4.0 * 0.1 / 30.0 ** 2 * S.gy * (30 - S.gy), 0.0)                              # a parabolic inlet
    assert not q.time_dependent() and q.space_dependent()
    assert nt.DynamicValue(S.gx) == nt.DynamicValue(S.gx) and hash(nt.DynamicValue(S.gx)) == hash(nt.DynamicValue(S.gx))


def _channel_sim(inlet, outlet):
    from sailfish_amd.lb_single import LBFluidSim
    from sailfish_amd.subdomain import Subdomain2D

    class Channel(Subdomain2D):
        def boundary_conditions(self, hx, hy):
            wall = (hy == 0) | (hy == self.gy - 1)
            self.set_node(wall, nt.NTFullBBWall)
            self.set_node((hx == 0) & ~wall, inlet(self))
            self.set_node((hx == self.gx - 1) & ~wall, outlet(self))

        def initial_conditions(self, sim, hx, hy):
            sim.rho[:] = 1.0

    class ChannelSim(LBFluidSim):
        subdomain = Channel
    return ChannelSim


def test_encoder_tables_for_constant_and_time_dependent_values():
    """A parabolic velocity inlet (position only): one table entry per distinct value, exactly what the same profile
    given as a per-node array produces.  A pulsating density outlet (time only): ONE entry, rewritten per step.  A value
    of both: an entry per node."""
    import sympy
    S = sym.S
    H = 14
    prof = lambda y: 4.0 * 0.05 / (H - 1.0) ** 2 * y * ((H - 1.0) - y)           # noqa: E731
    inlet = lambda sd: nt.NTEquilibriumVelocity(nt.DynamicValue(prof(S.gy), 0.0))      # noqa: E731
    outlet = lambda sd: nt.NTEquilibriumDensity(nt.DynamicValue(1.0 + 0.002 * sympy.sin(S.time * 0.1)))   # noqa: E731
    _, _, runners = _host.build_runners(_channel_sim(inlet, outlet), 2, LBGeometry2D, dict(lat_nx=24, lat_ny=H, visc=0.05))
    r = runners[0]
    r._init_geometry()
    enc = r._subdomain._encoder
    assert r.config.time_dependence and r.config.space_dependence and enc.time_dependent
    ups = enc.dynamic_updates(5)
    assert len(ups) == 1 and len(ups[0][1]) == 1 and abs(ups[0][1][0] - (1.0 + 0.002 * np.sin(0.5))) < 1e-15
    table = np.array(enc._geo_params)
    # the velocity entries: (vx, vy) pairs of the distinct values of the profile at y = 1 .. H - 2 (symmetric: half of them)
    want = set(round(prof(float(y)), 14) for y in range(1, H - 1))
    got = set(round(float(x), 14) for x in table[:-1][::2] if x != 0.0)
    assert want <= got and len(got) <= len(want) + 1       # (+ the ghost nodes y = -1, H the selection of the inlet includes)
    # what every inlet node's code points at: the profile at ITS height (reference Subdomain.get_param semantics)
    for y in range(1, H - 1):
        vx, vy = enc.get_param((1, y + 1), 2)       # array position: one ghost layer
        assert abs(vx - prof(float(y))) < 1e-15 and vy == 0.0
    both = lambda sd: nt.NTEquilibriumVelocity(nt.DynamicValue(prof(S.gy) * sympy.sin(S.time * 0.1), 0.0))    # noqa: E731
    _, _, r3 = _host.build_runners(_channel_sim(both, outlet), 2, LBGeometry2D, dict(lat_nx=24, lat_ny=H, visc=0.05))
    r3[0]._init_geometry()
    ups = r3[0]._subdomain._encoder.dynamic_updates(3)
    assert sorted(len(u[1]) for u in ups) == [1, 2 * H]      # H - 2 real inlet nodes + the two ghost nodes of the selection, (vx, vy) each


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
def test_pulsating_channel_through_the_runner_equals_the_oracle_twin(pattern):
    """The product's runner with the CPU test backend (kernels by the oracle, update_node_params before every step)
    against the oracle twin stepping the same geometry by hand: same populations, and the flow really follows the
    oscillating pressure (it speeds up while the pressure difference is positive and slows down once it has changed sign)."""
    import sympy
    from tests._oracle_backend import OracleBackend
    from tests._oracle_group import OracleGroup
    from sailfish_amd import util
    from sailfish_amd.controller import LBSimulationController
    S = sym.S
    amp, om = 0.004, 2 * np.pi / 60.0
    inlet = lambda sd: nt.NTEquilibriumDensity(nt.DynamicValue(1.0 + amp * sympy.sin(S.time * om)))     # noqa: E731
    outlet = lambda sd: nt.NTEquilibriumDensity(nt.DynamicValue(1.0 - amp * sympy.sin(S.time * om)))    # noqa: E731
    sim_cls = _channel_sim(inlet, outlet)
    cfg = dict(lat_nx=20, lat_ny=11, visc=0.08, access_pattern=pattern)
    means = []
    for steps in (16, 46):
        og = OracleGroup(sim_cls, 2, 'EqualSubdomainsGeometry2D', cfg)
        og.run(steps, save_last=True)
        old = util.get_backends
        util.get_backends = lambda backends=('hip',): iter([OracleBackend])
        try:
            ctrl = LBSimulationController(sim_cls, LBGeometry2D, default_config=dict(cfg, max_iters=steps, quiet=True,
                                                                                      perf_stats_every=0, backends='oracle_test'))
            ctrl.run(ignore_cmdline=True)
        finally:
            util.get_backends = old
        r = ctrl.runners[0]
        f = r._debug_get_dist()[(slice(None),) + tuple(r._spec._nonghost_slice)]
        fo = og.merged('dist')
        m = np.isfinite(fo)
        assert np.array_equal(f[m], fo[m])
        means.append(float(np.nanmean(r._sim.vx[1:-1, 2:-2])))
    assert means[0] > 5e-4 and means[1] < 0.6 * means[0], means       # the flow follows the pressure: it slows down again


@pytest.mark.parametrize('drive', ['force', 'pressure'])
def test_pulsatile_example_through_the_runner_equals_the_oracle_twin(drive):
    """examples/poiseuille_pulsatile.py: a body force resp. a pressure difference a0 sin(t).  The product's runner with the
    CPU test backend (set_body_force / update_node_params before every step) == the oracle twin, bit for bit; and the force
    the module sees really is the expression's value at the step's time."""
    from tests._oracle_backend import OracleBackend
    from tests._oracle_group import OracleGroup
    from sailfish_amd import util
    from sailfish_amd.controller import LBSimulationController
    sim_cls = _host.load_sim_class('poiseuille_pulsatile', 'PulsatileSim')
    cfg = dict(lat_nx=16, lat_ny=12, visc=0.05, drive=drive, horizontal=True, access_pattern='AA', dt_per_lattice_time_unit=0.05,
               wall='fullbb', stationary=False)
    steps = 33
    og = OracleGroup(sim_cls, 2, 'EqualSubdomainsGeometry2D', cfg)
    og.run(steps, save_last=True)
    old = util.get_backends
    util.get_backends = lambda backends=('hip',): iter([OracleBackend])
    try:
        ctrl = LBSimulationController(sim_cls, LBGeometry2D, default_config=dict(cfg, max_iters=steps, quiet=True, perf_stats_every=0,
                                                                                  backends='oracle_test'))
        ctrl.run(ignore_cmdline=True)
    finally:
        util.get_backends = old
    r = ctrl.runners[0]
    assert r._time_dependent()
    f = r._debug_get_dist()[(slice(None),) + tuple(r._spec._nonghost_slice)]
    fo = og.merged('dist')
    m = np.isfinite(fo)
    assert np.array_equal(f[m], fo[m])
    if drive == 'force':
        a0 = r._subdomain.max_v * 8.0 * 0.05 / r._subdomain.channel_width(r.config) ** 2
        assert abs(r._sim.body_force_at(10)[0] - a0 * np.sin(10 * 0.05)) < 1e-18 and r._sim.body_force_at(10)[1] == 0.0
        assert abs(r.module.desc.accel[0] - a0 * np.sin((steps - 1) * 0.05)) < 1e-18       # what the last step's launch took


def test_forces_that_depend_on_position_are_refused():
    from sailfish_amd.lb_base import LBForcedSim
    from sailfish_amd.lb_single import LBFluidSim

    class Sim(LBFluidSim, LBForcedSim):
        pass
    s = Sim(_host.make_config(2))
    with pytest.raises(NotImplementedError):
        s.add_body_force(nt.DynamicValue(1e-5 * sym.S.gy, 0.0))
