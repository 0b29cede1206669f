"""--regularized and --subgrid=les-smagorinsky (reference lb_single.py:27-42, relaxation_common.mako:166-237): the CPU
oracle against values composed from the reference's own sympy objects (tests/golden/arith_reg_les_*.npz,
tools/capture_goldens.py turbulence_goldens: bgk_equilibrium, ex_flux / ex_eq_flux, reglb_flux_tensor,
guo_external_force + its prefactor at the node's tau0), with and without a body force, two viscosities, two
Smagorinsky constants.  f64 oracle ~1e-13 (2e-12 where the subgrid model's square roots sit on a small argument), f32
oracle within the north-star tolerance 1e-6."""
import os

import numpy as np
import pytest

from oracle import oracle
from sailfish_amd import hipabi, sym

GRIDS = {'D2Q9': sym.D2Q9, 'D3Q19': sym.D3Q19}
TOL = {8: 2e-12, 4: 2e-6}


@pytest.fixture(scope='module', params=['D2Q9', 'D3Q19'])
def gold(request, golden_dir):
    return GRIDS[request.param], np.load(os.path.join(golden_dir, 'arith_reg_les_%s.npz' % request.param))


def desc_for(grid, precision, visc, csmag, accel, regularized, subgrid, **more):
    kw = dict(lattice=grid.slf_id, model=hipabi.SLF_BGK, precision=precision, access_pattern=hipabi.SLF_AB,
              lat_nx=4, lat_ny=4, lat_nz=4 if grid.dim == 3 else 1, arr_nx=4, arr_ny=4, arr_nz=4 if grid.dim == 3 else 1,
              tau=sym.relaxation_time(visc), visc=visc, mrt_rates=sym.mrt_rates(grid, visc),
              regularized=int(regularized), subgrid=hipabi.SLF_SUBGRID_LES_SMAGORINSKY if subgrid else hipabi.SLF_SUBGRID_NONE,
              smagorinsky_const=float(csmag))
    if accel is not None:
        kw['has_force'] = 1
        kw['accel'] = list(accel) + [0.0] * (3 - len(accel))
    kw.update(more)
    return hipabi.make_desc(**kw)


CASES = [('reg_post', True, False), ('les_post', False, True), ('reg_les_post', True, True)]


@pytest.mark.parametrize('precision', [8, 4])
@pytest.mark.parametrize('name,regularized,subgrid', CASES)
def test_oracle_against_the_reference_composition(gold, precision, name, regularized, subgrid):
    grid, G = gold
    n = len(G['f'])
    for vi, nu in enumerate(G['visc']):
        for ci, c in enumerate(G['smagorinsky_const']):
            for forced in (0, 1):
                for k in range(n):
                    d = desc_for(grid, precision, float(nu), float(c), G['accel'][k] if forced else None, regularized, subgrid)
                    f, rho, v = oracle.node_update(d, hipabi.SLF_NK_FLUID, 0, None, G['f'][k], precision)
                    err = np.max(np.abs(f - G[name][vi, ci, forced, k]))
                    assert err < TOL[precision], (name, vi, ci, forced, k, err)
                    assert np.max(np.abs(v[:grid.dim] - G['out_v'][forced, k])) < TOL[precision]


def test_the_options_change_something_and_nothing_else(gold):
    """Both options off: the plain BGK collision, to the bit; each one on: a different result (the fixtures are not the
    plain collision in disguise); the subgrid model raises the relaxation time, never lowers it."""
    grid, G = gold
    nu, c = float(G['visc'][0]), float(G['smagorinsky_const'][1])
    plain = hipabi.make_desc(lattice=grid.slf_id, model=hipabi.SLF_BGK, precision=8, access_pattern=hipabi.SLF_AB, lat_nx=4,
                             lat_ny=4, lat_nz=4 if grid.dim == 3 else 1, arr_nx=4, arr_ny=4, arr_nz=4 if grid.dim == 3 else 1,
                             tau=sym.relaxation_time(nu), visc=nu, mrt_rates=sym.mrt_rates(grid, nu))
    for k in range(4):
        f0, _, _ = oracle.node_update(plain, hipabi.SLF_NK_FLUID, 0, None, G['f'][k], 8)
        f1, _, _ = oracle.node_update(desc_for(grid, 8, nu, c, None, False, False), hipabi.SLF_NK_FLUID, 0, None, G['f'][k], 8)
        assert np.array_equal(f0, f1)
        for reg, les in ((True, False), (False, True)):
            f2, _, _ = oracle.node_update(desc_for(grid, 8, nu, c, None, reg, les), hipabi.SLF_NK_FLUID, 0, None, G['f'][k], 8)
            assert np.max(np.abs(f2 - f0)) > 1e-6
    assert np.all(G['les_tau'][0] > sym.relaxation_time(nu))


def test_conserved_moments_survive_both_options(gold):
    """Neither option touches density or momentum (the projection has no zeroth or first moment; the subgrid model only
    rescales the relaxation): sum f and sum e f of the collided populations equal those of the populations that came in."""
    grid, G = gold
    nu, c = float(G['visc'][1]), float(G['smagorinsky_const'][0])
    e = np.array([[int(x) for x in b] for b in grid.basis], dtype=np.float64)
    for k in range(6):
        for reg, les in ((True, False), (False, True), (True, True)):
            f, _, _ = oracle.node_update(desc_for(grid, 8, nu, c, None, reg, les), hipabi.SLF_NK_FLUID, 0, None, G['f'][k], 8)
            assert abs(f.sum() - G['f'][k].sum()) < 1e-13
            assert np.max(np.abs(f @ e - G['f'][k] @ e)) < 1e-13
