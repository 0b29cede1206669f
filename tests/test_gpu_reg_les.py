"""--regularized and --subgrid=les-smagorinsky on the GPU (reference lb_single.py:27-42, relaxation_common.mako:166-237):
(i) the HIP per-node kernels against the values composed from the reference's own sympy objects
(tests/golden/arith_reg_les_*.npz) directly -- a periodic box whose nodes all carry one fixture state is invariant under
streaming, so one step leaves every node with the fixture's post-collision state; (ii) HIP == oracle bit for bit on a
periodic box, a lid-driven cavity and a force-driven channel, both access patterns, both precisions; (iii) what the
library refuses."""
import os

import numpy as np
import pytest

from sailfish_amd import hipabi, sym
from sailfish_amd.box import make_box_desc
from tests import _geometry as geo
from tests.test_gpu_golden import GRIDS, _check, _one_step
from tests.test_gpu_parity import RTOL, _run_pair

pytestmark = pytest.mark.gpu
TOL = {'double': 2e-12, 'single': 2e-6}
CASES = [('reg_post', True, False), ('les_post', False, True), ('reg_les_post', True, True)]


@pytest.fixture(scope='module')
def backend():
    from sailfish_amd.backend_hip import HIPBackend

    class Opt(object):
        pass
    return HIPBackend(Opt(), 0)


@pytest.mark.parametrize('fused', [1, 0], ids=['in_sweep_wrap', 'ghost_pbc'])
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_kernels_against_the_reference_composition(backend, golden_dir, name, precision, fused):
    grid, size = GRIDS[name]
    G = np.load(os.path.join(golden_dir, 'arith_reg_les_%s.npz' % name))
    tol = TOL[precision]
    for k in (0, 5, 11, 23):
        for vi, nu in enumerate(G['visc']):
            for ci, c in enumerate(G['smagorinsky_const']):
                for forced in (0, 1):
                    for fixture, reg, les in CASES:
                        res = _one_step(backend, grid, size, G['f'][k], precision, fused, model='bgk', visc=float(nu),
                                        accel=list(G['accel'][k]) if forced else None, regularized=reg, subgrid=les,
                                        smagorinsky_const=float(c))
                        _check(res, G[fixture][vi, ci, forced, k], tol)
                        assert max(abs(a - b) for a, b in zip(res[1][1], G['out_v'][forced, k])) < tol


@pytest.mark.parametrize('grid,size', [(sym.D2Q9, (70, 11)), (sym.D3Q19, (70, 6, 5))])
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('precision', ['single', 'double'])
@pytest.mark.parametrize('reg,les', [(True, False), (False, True), (True, True)])
def test_periodic_box_equals_the_oracle(backend, grid, size, pattern, precision, reg, les):
    r = _run_pair(backend, grid, size, 21, (True, True, True), model='bgk', precision=precision, access_pattern=pattern,
                  visc=0.002, periodic_fused=[1, 1, 1], regularized=reg, subgrid=les, smagorinsky_const=0.14)
    assert r['dist_exact'], r
    assert r['rho_err'] < RTOL and r['v_err'] < RTOL, r


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('reg,les', [(True, False), (False, True)])
def test_cavity_and_forced_channel_equal_the_oracle(backend, pattern, reg, les):
    """Node map, boundary-condition nodes (the regularized lid does its own regularisation first, then the collision's),
    bounce-back walls, a body force."""
    size = (40, 12, 9)
    r = _run_pair(backend, sym.D3Q19, size, 30, (False, False, False), node_map_fn=geo.cavity_3d, init='rest', model='bgk',
                  precision='single', access_pattern=pattern, visc=0.004, fluid_only=False, type_kind=geo.TYPE_KIND,
                  nt_bits=geo.NT_BITS, node_params=[0.05, 0.0, 0.0], regularized=reg, subgrid=les, smagorinsky_const=0.1)
    assert r['dist_exact'], r
    r = _run_pair(backend, sym.D3Q19, (36, 10, 6), 40, (True, False, True), node_map_fn=geo.channel_3d_fullbb, init='rest',
                  model='bgk', precision='single', access_pattern=pattern, visc=0.003, fluid_only=False, type_kind=geo.TYPE_KIND,
                  nt_bits=geo.NT_BITS, accel=[2e-5, 0.0, 0.0], periodic_fused=[1, 0, 1], regularized=reg, subgrid=les,
                  smagorinsky_const=0.1)
    assert r['dist_exact'], r


def test_the_subgrid_model_damps_what_plain_bgk_does_not(backend):
    """At a viscosity where plain BGK is close to its stability limit the Smagorinsky closure raises the local relaxation
    time where the strain is large: a sheared periodic box keeps finite populations and loses kinetic energy faster."""
    from sailfish_amd.box import BoxSim
    from tests._oracle_box import synthetic_fields
    size = (64, 64)
    energy = {}
    for les in (False, True):
        desc = make_box_desc(sym.D2Q9, size, model='bgk', precision='single', access_pattern='AB', visc=2e-4,
                             periodic_fused=[1, 1, 1], subgrid=les, smagorinsky_const=0.17)
        s = BoxSim(backend, desc, periodic=(True, True, True))
        rho, v = synthetic_fields(size, 2)
        s.set_fields(rho, [2.0 * c for c in v])
        s.initial_conditions()
        s.run(400, save_last=True)
        _, vv = s.fetch_fields()
        energy[les] = float(sum((s.real_view(c).astype(np.float64) ** 2).sum() for c in vv))
        assert np.isfinite(energy[les])
        s.release()
    assert energy[True] < energy[False]


def test_refusals(backend):
    """MRT (the reference's MRT relaxation never calls the preamble that implements the options: it would ignore them
    silently), the exact difference method, --minimize_roundoff."""
    for kw in (dict(model='mrt'), dict(model='bgk', incompressible=hipabi.SLF_DENSITY_ROUNDOFF)):
        desc = make_box_desc(sym.D3Q19, (16, 4, 4), precision='single', access_pattern='AB', visc=0.01,
                             periodic_fused=[1, 1, 1], regularized=True, **kw)
        with pytest.raises(backend.FatalError, match='regularized / subgrid'):
            backend.build(desc)
    desc = make_box_desc(sym.D3Q19, (16, 4, 4), precision='single', access_pattern='AB', visc=0.01, periodic_fused=[1, 1, 1],
                         accel=[1e-5, 0, 0], subgrid=True)
    desc.force_implementation = hipabi.SLF_FORCE_EDM
    with pytest.raises(backend.FatalError, match='regularized / subgrid'):
        backend.build(desc)
    desc = make_box_desc(sym.D3Q19, (16, 4, 4), precision='single', access_pattern='AB', visc=0.01, periodic_fused=[1, 1, 1],
                         subgrid=True, smagorinsky_const=0.0)
    with pytest.raises(backend.FatalError, match='smagorinsky_const'):
        backend.build(desc)
