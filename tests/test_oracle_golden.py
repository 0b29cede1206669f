"""Pin the CPU oracle (oracle/lbm_oracle.c) against values obtained by evaluating
the reference's own sympy expressions (tests/golden/arith_*.npz, produced by
tools/capture_goldens.py from sailfish/sym.py, sym_equilibrium.py, sym_force.py).

f64 oracle: agreement to ~1e-13 (different but equivalent operation order);
f32 oracle: agreement within the north-star tolerance 1e-6 (relative to the
population scale w_i * rho).
"""
import os

import numpy as np
import pytest

from oracle import oracle
from sailfish_amd import hipabi, sym

GRIDS = {'D2Q9': sym.D2Q9, 'D3Q19': sym.D3Q19}
TOL = {8: 2e-13, 4: 1e-6}


@pytest.fixture(scope='module', params=['D2Q9', 'D3Q19'])
def gold(request, golden_dir):
    g = GRIDS[request.param]
    return g, np.load(os.path.join(golden_dir, 'arith_%s.npz' % request.param))


def _desc(grid, precision, visc=1.0 / 6.0, model='bgk', accel=None, incompressible=False, edm=False):
    kw = dict(lattice=grid.slf_id, model=hipabi.SLF_MRT if model == 'mrt' else hipabi.SLF_BGK,
              precision=precision, access_pattern=hipabi.SLF_AB,
              lat_nx=4, lat_ny=4, lat_nz=4 if grid.dim == 3 else 1,
              arr_nx=4, arr_ny=4, arr_nz=4 if grid.dim == 3 else 1,
              tau=sym.relaxation_time(visc), visc=visc, mrt_rates=sym.mrt_rates(grid, visc),
              incompressible=int(incompressible))
    if accel is not None:
        kw['has_force'] = 1
        kw['accel'] = list(accel) + [0.0] * (3 - len(accel))
    if edm:
        kw['force_implementation'] = hipabi.SLF_FORCE_EDM
    return hipabi.make_desc(**kw)


def _close(a, b, tol, scale=1.0):
    err = np.max(np.abs(np.asarray(a) - np.asarray(b))) / scale
    assert err < tol, 'max err %.3e (tol %.1e)' % (err, tol)


@pytest.mark.parametrize('precision', [8, 4])
def test_equilibrium(gold, precision):
    grid, G = gold
    for inc in (0, 1):
        for k in range(len(G['rho'])):
            got = oracle.node_feq(grid.slf_id, G['rho'][k], G['v'][k], bool(inc), precision)
            _close(got, G['feq_inc%d' % inc][k], TOL[precision])
            # sum feq = rho, sum e feq = rho0 u  (reference tests/sym_equilibrium.py:93-122)
            if precision == 8:
                assert abs(got.sum() - G['rho'][k]) < 1e-13


@pytest.mark.parametrize('precision', [8, 4])
def test_moments(gold, precision):
    grid, G = gold
    for k in range(len(G['rho'])):
        rho, v = oracle.node_macro(grid.slf_id, G['f'][k], False, precision)
        _close(rho, G['mom_rho'][k], TOL[precision])
        _close(v[:grid.dim], G['mom_v'][k], TOL[precision])
        rho, m = oracle.node_macro(grid.slf_id, G['f'][k], True, precision)
        _close(m[:grid.dim], G['mom_momentum'][k], TOL[precision])


@pytest.mark.parametrize('precision', [8, 4])
def test_bgk_collision(gold, precision):
    grid, G = gold
    for a, nu in enumerate(G['bgk_visc']):
        d = _desc(grid, precision, visc=float(nu))
        for k in range(len(G['rho'])):
            f, rho, v = oracle.node_update(d, hipabi.SLF_NK_FLUID, 0, None, G['f'][k], precision)
            _close(f, G['bgk_post'][a, k], TOL[precision])


@pytest.mark.parametrize('precision', [8, 4])
def test_bgk_guo_force(gold, precision):
    grid, G = gold
    nu = float(G['guo_visc'][0])
    for k in range(len(G['rho'])):
        d = _desc(grid, precision, visc=nu, accel=G['accel'][k])
        f, rho, v = oracle.node_update(d, hipabi.SLF_NK_FLUID, 0, None, G['f'][k], precision)
        _close(f, G['guo_post'][k], TOL[precision])
        _close(v[:grid.dim], G['guo_out_v'][k], TOL[precision])


@pytest.mark.parametrize('precision', [8, 4])
def test_bgk_edm_force(gold, precision):
    """Exact difference method (reference relaxation_common.mako:66-73, sym_force.edm_shift_velocity)."""
    grid, G = gold
    nu = float(G['guo_visc'][0])
    for k in range(len(G['rho'])):
        d = _desc(grid, precision, visc=nu, accel=G['accel'][k], edm=True)
        f, rho, v = oracle.node_update(d, hipabi.SLF_NK_FLUID, 0, None, G['f'][k], precision)
        _close(f, G['edm_post'][k], TOL[precision])
        _close(v[:grid.dim], G['guo_out_v'][k], TOL[precision])      # same output velocity u + a/2
        assert np.max(np.abs(G['edm_post'][k] - G['guo_post'][k])) > 0   # the two schemes do differ


@pytest.mark.parametrize('precision', [8, 4])
def test_mrt_collision(gold, precision):
    grid, G = gold
    for a, nu in enumerate(G['bgk_visc']):
        d = _desc(grid, precision, visc=float(nu), model='mrt')
        for k in range(len(G['rho'])):
            f, rho, v = oracle.node_update(d, hipabi.SLF_NK_FLUID, 0, None, G['f'][k], precision)
            # the f32 MRT round trip multiplies by integers up to 30 -> allow 4e-6
            _close(f, G['mrt_post'][a, k], TOL[precision] * (4 if precision == 4 else 50))


@pytest.mark.parametrize('precision', [8, 4])
def test_mrt_pair_form_equals_matrix_form(gold, precision):
    """D3Q19: the moment transform evaluated through the pairs of opposite directions (what the oracle and the HIP
    kernels run) against the row-by-row products with the integer matrix of sym.py:331-378: the same numbers up to the
    rounding of a differently ordered sum, and both within the tolerance of the reference's sympy values."""
    grid, G = gold
    if grid.Q != 19:
        pytest.skip('D2Q9 keeps the matrix form')
    worst = 0.0
    try:
        for a, nu in enumerate(G['bgk_visc']):
            d = _desc(grid, precision, visc=float(nu), model='mrt')
            for k in range(len(G['rho'])):
                oracle.set_mrt_form(False, precision)
                f_pair, _, _ = oracle.node_update(d, hipabi.SLF_NK_FLUID, 0, None, G['f'][k], precision)
                oracle.set_mrt_form(True, precision)
                f_mat, _, _ = oracle.node_update(d, hipabi.SLF_NK_FLUID, 0, None, G['f'][k], precision)
                worst = max(worst, float(np.max(np.abs(f_pair - f_mat))))
                _close(f_mat, G['mrt_post'][a, k], TOL[precision] * (4 if precision == 4 else 50))
                _close(f_pair, G['mrt_post'][a, k], TOL[precision] * (4 if precision == 4 else 50))
    finally:
        oracle.set_mrt_form(False, precision)
    assert 0.0 < worst < (2e-6 if precision == 4 else 2e-14)       # different rounding, same transform
    # the pair form is the more accurate of the two: fewer, smaller intermediate sums
    print('max |pair - matrix| = %.3e' % worst)


@pytest.mark.parametrize('precision', [8, 4])
def test_mrt_collision_with_body_force(gold, precision):
    """Moment-space forcing (reference relaxation_mrt.mako:10-27, 45-46, 91: half of sym_force.accel_vector in the
    momentum moments before the equilibrium, half after the relaxation; output velocity u + a / 2)."""
    grid, G = gold
    nu = float(G['mrt_force_visc'][0])
    for k in range(len(G['rho'])):
        d = _desc(grid, precision, visc=nu, model='mrt', accel=G['accel'][k])
        f, rho, v = oracle.node_update(d, hipabi.SLF_NK_FLUID, 0, None, G['f'][k], precision)
        _close(f, G['mrt_force_post'][k], TOL[precision] * (4 if precision == 4 else 50))
        _close(v[:grid.dim], G['guo_out_v'][k], TOL[precision])
        assert np.max(np.abs(G['mrt_force_post'][k] - G['mrt_post'][len(G['bgk_visc']) // 2, k])) > 0


def _no_relax(grid, precision):
    d = _desc(grid, precision)
    d.relaxation_enabled = 0
    return d


@pytest.mark.parametrize('precision', [8, 4])
def test_regularized_velocity_bc(gold, precision):
    grid, G = gold
    d = _no_relax(grid, precision)
    for o in range(1, 2 * grid.dim + 1):
        for k in range(len(G['rho'])):
            f, rho, v = oracle.node_update(d, hipabi.SLF_NK_REGULARIZED_VELOCITY, o, G['bc_v'][k], G['f'][k],
                                           precision)
            _close(rho, G['regvel_rho'][o - 1, k], TOL[precision] * 2)
            _close(f, G['regvel_post'][o - 1, k], TOL[precision] * 2)
            _close(v[:grid.dim], G['bc_v'][k], 1e-7)


@pytest.mark.parametrize('precision', [8, 4])
def test_equilibrium_density_bc(gold, precision):
    grid, G = gold
    d = _no_relax(grid, precision)
    for o in range(1, 2 * grid.dim + 1):
        for k in range(len(G['rho'])):
            f, rho, v = oracle.node_update(d, hipabi.SLF_NK_EQUILIBRIUM_DENSITY, o, [G['bc_rho'][k]], G['f'][k],
                                           precision)
            _close(rho, G['bc_rho'][k], 1e-7)
            # v = -(n)(rho_s - par_rho)/par_rho: the f32 cancellation error is ~eps*rho/par_rho
            _close(v[:grid.dim], G['eqdens_v'][o - 1, k], TOL[precision] * 2)
            _close(f, G['eqdens_post'][o - 1, k], TOL[precision] * 2)


def test_full_bounce_back(gold):
    grid, G = gold
    d = _desc(grid, 8)
    f0 = G['f'][0]
    f, rho, v = oracle.node_update(d, hipabi.SLF_NK_FULL_BB, 0, None, f0, 8)
    for i in range(grid.Q):
        assert f[i] == f0[grid.idx_opposite[i]]


@pytest.mark.parametrize('precision', [8, 4])
def test_zouhe_velocity_bc(gold, precision):
    """NTZouHeVelocity: rho from the known populations, non-equilibrium bounce-back, tangential momentum
    fix-up (reference boundary.mako:343-382, sym.zouhe_fixup) -- expected values from the reference's sympy."""
    grid, G = gold
    d = _no_relax(grid, precision)
    for o in range(1, 2 * grid.dim + 1):
        for k in range(len(G['rho'])):
            f, rho, v = oracle.node_update(d, hipabi.SLF_NK_ZOUHE_VELOCITY, o, G['bc_v'][k], G['f'][k], precision)
            _close(rho, G['regvel_rho'][o - 1, k], TOL[precision] * 2)
            _close(f, G['zouhe_vel_post'][o - 1, k], TOL[precision] * 2)
            # the node now carries exactly the imposed momentum
            e = grid.basis_array
            for a in range(grid.dim):
                assert abs(sum(e[i][a] * f[i] for i in range(grid.Q)) - rho * G['bc_v'][k][a]) < TOL[precision] * 4


@pytest.mark.parametrize('precision', [8, 4])
def test_zouhe_density_bc(gold, precision):
    grid, G = gold
    d = _no_relax(grid, precision)
    for o in range(1, 2 * grid.dim + 1):
        for k in range(len(G['rho'])):
            f, rho, v = oracle.node_update(d, hipabi.SLF_NK_ZOUHE_DENSITY, o, [G['bc_rho'][k]], G['f'][k], precision)
            _close(rho, G['bc_rho'][k], 1e-7)
            _close(f, G['zouhe_dens_post'][o - 1, k], TOL[precision] * 2)
            _close(v[:grid.dim], G['zouhe_dens_v'][o - 1, k], TOL[precision] * 2)


@pytest.mark.parametrize('precision', [8, 4])
def test_regularized_density_bc(gold, precision):
    grid, G = gold
    d = _no_relax(grid, precision)
    for o in range(1, 2 * grid.dim + 1):
        for k in range(len(G['rho'])):
            f, rho, v = oracle.node_update(d, hipabi.SLF_NK_REGULARIZED_DENSITY, o, [G['bc_rho'][k]], G['f'][k],
                                           precision)
            _close(rho, G['bc_rho'][k], 1e-7)
            _close(v[:grid.dim], G['eqdens_v'][o - 1, k], TOL[precision] * 2)
            _close(f, G['regdens_post'][o - 1, k], TOL[precision] * 2)
