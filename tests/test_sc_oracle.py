"""Binary Shan-Chen model, CPU side: the oracle's force against the template formula
(reference templates/shan_chen.mako:29-84 with the pseudopotentials of sym.py:896-908) and the
model-level invariants (per-component mass, total momentum, AB == AA, ghost-PBC == in-sweep wrap)."""
import numpy as np
import pytest

from oracle import oracle
from sailfish_amd import sym
from tests import _host, _sc
from tests._oracle_group import OracleSCSubdomain


@pytest.mark.parametrize('grid', [sym.D2Q9, sym.D3Q19])
@pytest.mark.parametrize('potential', [0, 1])
def test_force_formula(grid, potential):
    rng = np.random.RandomState(3)
    e = grid.basis_array
    w = grid.weights_float
    psi = (lambda r: r) if potential == 0 else (lambda r: 1.0 - np.exp(-r))
    for _ in range(20):
        neigh = rng.uniform(0.5, 1.5, grid.Q)
        rho = rng.uniform(0.5, 1.5)
        G = rng.uniform(0.5, 2.0)
        ref = np.zeros(3)
        for i in range(1, grid.Q):
            ref[:grid.dim] += w[i] * e[i] * psi(neigh[i])
        ref *= -G * psi(rho)
        got = oracle.sc_force_node(grid.slf_id, potential, G, rho, neigh, precision=8)
        assert np.max(np.abs(got - ref)) < 1e-14
        got32 = oracle.sc_force_node(grid.slf_id, potential, G, rho, neigh, precision=4)
        assert np.max(np.abs(got32 - ref)) < 2e-6


def _run(dim, size, steps, **kw):
    sim_cls, geo = _sc.make_sim(dim)
    cfg_, specs, runners = _host.build_runners(sim_cls, dim, geo, _sc.config(dim, size, **kw))
    s = OracleSCSubdomain(runners[0])
    f0 = [s.real(d).astype(np.float64).copy() for d in s.current()]
    s.run(steps)
    return s, f0


@pytest.mark.parametrize('dim,size', [(2, (18, 14)), (3, (10, 8, 7))])
def test_conservation(dim, size):
    s, f0 = _run(dim, size, 20, pattern='AB', precision='double')
    f1 = [s.real(d) for d in s.current()]
    grid = s.runner._sim.grid
    for a, b in zip(f0, f1):        # each component keeps its mass
        assert abs(a.sum() - b.sum()) < 1e-10
    e = grid.basis_array
    for d in range(dim):            # the coupling is momentum conserving (G12 = G21)
        m0 = sum(e[i][d] * (f0[0][i].sum() + f0[1][i].sum()) for i in range(grid.Q))
        m1 = sum(e[i][d] * (f1[0][i].sum() + f1[1][i].sum()) for i in range(grid.Q))
        assert abs(m0 - m1) < 1e-9


@pytest.mark.parametrize('dim,size', [(2, (18, 14)), (3, (10, 8, 7))])
def test_ab_aa_and_fused_equivalence(dim, size):
    res = {}
    for pattern in ('AB', 'AA'):
        for fused in (True, False):
            s, _ = _run(dim, size, 10, pattern=pattern, fused=fused)
            res[(pattern, fused)] = [s.real(d).copy() for d in s.current()] + [s.real(s.rho).copy()]
    ref = res[('AB', True)]
    for k, v in res.items():
        for a, b in zip(ref, v):
            assert np.array_equal(a, b), k


def test_phase_separation_starts():
    """G12 = 1.2 is above the spinodal threshold: the contrast of the initial noise grows."""
    s, _ = _run(2, (32, 32), 400, pattern='AA')
    d = s.real(s.rho) - s.real(s.phi)
    assert np.abs(d).max() > 0.05


def _run_single(dim, size, steps, **kw):
    from tests._oracle_group import OracleSCSingle
    sim_cls, geo = _sc.make_single_sim(dim)
    cfg_, specs, runners = _host.build_runners(sim_cls, dim, geo, _sc.single_config(dim, size, **kw))
    s = OracleSCSingle(runners[0])
    m0 = s.real(s.current()).astype(np.float64).sum()
    s.run(steps)
    return s, m0


@pytest.mark.parametrize('dim,size', [(2, (18, 14)), (3, (10, 8, 7))])
def test_single_component_invariants(dim, size):
    res = {}
    for pattern in ('AB', 'AA'):
        for fused in (True, False):
            s, m0 = _run_single(dim, size, 12, pattern=pattern, fused=fused, potential='linear', G=-1.0)
            res[(pattern, fused)] = (s.real(s.current()).copy(), s.real(s.rho).copy(), s.real(s.v[0]).copy())
            assert abs(s.real(s.current()).astype(np.float64).sum() - m0) / m0 < 1e-6
    ref = res[('AB', True)]
    for k, v in res.items():
        for a, b in zip(ref, v):
            assert np.array_equal(a, b), k


def _group(dim, size, steps, single, nsub, axis, **kw):
    from tests._oracle_group import OracleNNGroup
    sim_cls, _ = (_sc.make_single_sim if single else _sc.make_sim)(dim)
    cfg = (_sc.single_config if single else _sc.config)(dim, size, **kw)
    if single:
        cfg.update(G=-1.2, sc_potential='linear')
    cfg.update(subdomains=nsub, conn_axis=axis)
    g = OracleNNGroup(sim_cls, dim, 'EqualSubdomainsGeometry%dD' % dim, cfg, single=single)
    g.run(steps)
    return g


@pytest.mark.parametrize('single', [False, True])
@pytest.mark.parametrize('dim,size,nsub,axis', [(2, (18, 12), 2, 'x'), (2, (18, 12), 3, 'y'), (3, (10, 8, 6), 2, 'z'),
                                                (3, (10, 8, 6), 2, 'x')])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
def test_one_vs_n_subdomains(single, dim, size, nsub, axis, pattern):
    """Macro-field halo + population halo of the non-local models: N subdomains == 1 subdomain, bit for bit
    (reference regtest/subdomains/binary_pbc.py checks 6 decimals)."""
    ref = _group(dim, size, 9, single, 1, axis, pattern=pattern)
    for fused in (True, False):
        g = _group(dim, size, 9, single, nsub, axis, pattern=pattern, fused=fused)
        assert np.array_equal(g.merged(lambda s: s.rho), ref.merged(lambda s: s.rho))
        if single:
            assert np.array_equal(g.merged(lambda s: s.current()), ref.merged(lambda s: s.current()))
        else:
            assert np.array_equal(g.merged(lambda s: s.phi), ref.merged(lambda s: s.phi))
            for k in (0, 1):
                assert np.array_equal(g.merged(lambda s: s.current()[k]), ref.merged(lambda s: s.current()[k]))


@pytest.mark.parametrize('dim,size', [(2, (16, 12)), (3, (8, 6, 6))])
def test_per_lattice_body_force(dim, size):
    """add_body_force(a, grid=k) accelerates lattice k only (reference relaxation_common.mako:9-36): in a
    periodic box the momentum of the mixture grows by  sum_k mass_k a_k  per step (the Shan-Chen coupling is
    momentum conserving), and a force on lattice 1 leaves the result unchanged when phi carries no mass...
    here: compared against the unforced run for the expected difference."""
    a0 = [2e-5, 0.0, -1e-5][:dim]
    a1 = [0.0, 3e-5, 1e-5][:dim]
    res = {}
    for key, (f0, f1) in {'none': (None, None), 'both': (a0, a1), 'only1': (None, a1)}.items():
        sim_cls, geo = _sc.make_forced_sim(dim, f0, f1)
        cfg_, specs, runners = _host.build_runners(sim_cls, dim, geo, _sc.config(dim, size, pattern='AB', precision='double'))
        s = OracleSCSubdomain(runners[0])
        assert list(s.desc.accel1)[:dim] == (list(f1) if f1 else [0.0] * dim)
        steps = 6
        s.run(steps)
        grid = s.runner._sim.grid
        e = grid.basis_array
        f = [s.real(d).astype(np.float64) for d in s.current()]
        mom = [[sum(e[i][d] * f[k][i].sum() for i in range(grid.Q)) for d in range(dim)] for k in (0, 1)]
        mass = [f[k].sum() for k in (0, 1)]
        res[key] = (np.array(mom[0]) + np.array(mom[1]), mass, steps)
    m_none, mass, steps = res['none']
    for key, (f0, f1) in (('both', (a0, a1)), ('only1', (None, a1))):
        gain = res[key][0] - m_none
        expect = steps * (mass[0] * np.array(f0 if f0 else [0.0] * dim) + mass[1] * np.array(f1))
        assert np.allclose(gain, expect, rtol=1e-6, atol=1e-10), (key, gain, expect)


# ---- pinned to the reference: fixtures composed from its own objects (tools/capture_goldens.py: shan_chen_goldens)
@pytest.mark.parametrize('grid', [sym.D2Q9, sym.D3Q19])
def test_pseudopotentials_match_reference(grid, golden_dir):
    """sym.SHAN_CHEN_POTENTIALS (sym.py:896-908) through the oracle's force routine: one neighbour carries psi."""
    import os
    g = np.load(os.path.join(golden_dir, 'shan_chen_%s.npz' % grid.__name__))
    w1 = grid.weights_float[1]
    for pot_id, pot in ((0, 'linear'), (1, 'classic')):
        for x, psi in zip(g['psi_in'], g['psi_' + pot]):
            neigh = np.zeros(grid.Q)
            neigh[1] = x                      # e_1 = +x; psi(0) = 0 for both potentials
            # F_x = -G psi(rho_loc) w_1 psi(x); with G = -1 and rho_loc chosen so that psi(rho_loc) = known
            out = oracle.sc_force_node(grid.slf_id, pot_id, -1.0, x, neigh, precision=8)
            assert abs(out[0] - psi * psi * w1) < 2e-14


@pytest.mark.parametrize('grid', [sym.D2Q9, sym.D3Q19])
@pytest.mark.parametrize('potential', ['linear', 'classic'])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
def test_node_update_matches_reference_composition(grid, potential, pattern, golden_dir):
    """One node of the binary Shan-Chen model through the oracle's whole-field kernels -- ShanChenPrepareMacroFields
    then ShanChenCollideAndPropagate0/1 -- against values composed from the reference's own sympy objects
    (pseudopotential, second-lattice equilibrium rho = rho0 = phi, common velocity, per-lattice Guo term) in f64.
    A 3^dim block of fluid nodes: the centre node carries the sample's populations, its neighbours the sample's
    rho / phi values."""
    import os
    from sailfish_amd import hipabi
    g = np.load(os.path.join(golden_dir, 'shan_chen_%s.npz' % grid.__name__))
    dim, Q = grid.dim, grid.Q
    e = grid.basis_array
    visc, tau_phi = float(g['visc'][0]), float(g['tau_phi'][0])
    n = len(g['f1'])
    lat = [5, 5, 5 if dim == 3 else 1]
    worst = 0.0
    for k in range(n):
        desc = hipabi.make_desc(lattice=grid.slf_id, model=hipabi.SLF_BGK, precision=8,
                                access_pattern=hipabi.SLF_AA if pattern == 'AA' else hipabi.SLF_AB,
                                lat_nx=5, lat_ny=5, lat_nz=lat[2], arr_nx=32, arr_ny=5, arr_nz=lat[2], fluid_only=1,
                                tau=sym.relaxation_time(visc), visc=visc, tau_phi=tau_phi,
                                simtype=hipabi.SLF_SIM_SHAN_CHEN_BINARY, sc_G=list(g['G'][k]),
                                sc_potential=0 if potential == 'linear' else 1,
                                accel=list(g['body_accel'][k, 0]) + [0.0] * (3 - dim),
                                accel1=list(g['body_accel'][k, 1]) + [0.0] * (3 - dim),
                                has_force=int(np.any(g['body_accel'][k] != 0.0)))
        o = oracle.OracleSim(desc)
        c = (2, 2, 2) if dim == 3 else (0, 2, 2)            # (z, y, x) of the centre node
        d1, d2 = o.new_dist(), o.new_dist()
        w = np.array(grid.weights_float)
        for d, f in ((d1, g['f1'][k]), (d2, g['f2'][k])):
            d[...] = w.reshape((Q,) + (1,) * 3)              # benign values everywhere
            d[(slice(None),) + c] = f
        rho, phi = o.new_field(1.0), o.new_field(1.0)
        v = [o.new_field(0.0) for _ in range(3)]
        # even AA step / AB: the node's own slots hold f_i
        o.sc_macro(1 if pattern == 'AA' else 0, None, d1, d2, rho, phi, v[0], v[1], v[2])
        assert abs(rho[c] - g['sc_rho'][k]) < 1e-14 and abs(phi[c] - g['sc_phi'][k]) < 1e-14
        for a in range(dim):
            assert abs(v[a][c] - g['sc_v'][k, a]) < 1e-15
        for i in range(1, Q):                                 # the neighbours' densities as in the sample
            p = (c[0] + (e[i][2] if dim == 3 else 0), c[1] + e[i][1], c[2] + e[i][0])
            rho[p], phi[p] = g['rho_nb'][k, i], g['phi_nb'][k, i]
        post = g['sc_post_' + potential][k]
        for l, (din, field) in enumerate(((d1, rho), (d2, phi))):
            dout = din if pattern == 'AA' else o.new_dist()
            o.sc_step(l, 1 if pattern == 'AA' else 0, None, din, dout, rho, phi, v[0], v[1], v[2])
            for i in range(Q):
                if pattern == 'AA':                           # in place, opposite slot
                    got = dout[(grid.idx_opposite[i],) + c]
                else:                                         # pushed to x + e_i
                    p = (c[0] + (e[i][2] if dim == 3 else 0), c[1] + e[i][1], c[2] + e[i][0])
                    got = dout[(i,) + p]
                worst = max(worst, abs(got - post[l, i]))
    assert worst < 1e-13, worst
