"""HIP kernels against the reference-derived fixtures DIRECTLY, without the oracle in between.

A periodic box in which every node carries the same populations f is invariant under streaming: after one
two-copy step every node holds the post-collision state of f, and tests/golden/arith_*.npz has that state evaluated
from the reference's own sympy objects (tools/capture_goldens.py: sym_equilibrium.bgk_equilibrium,
sym_force.guo_external_force / edm_shift_velocity / accel_vector, grid.mrt_*).  f64 to 2e-13, f32 to the north-star
tolerance 1e-6 (absolute, on populations of order w_i)."""
import os

import numpy as np
import pytest

from sailfish_amd import hipabi, sym
from sailfish_amd.box import BoxSim, make_box_desc

pytestmark = pytest.mark.gpu
GRIDS = {'D2Q9': (sym.D2Q9, (66, 5)), 'D3Q19': (sym.D3Q19, (64, 4, 3))}
TOL = {'double': 2e-13, 'single': 1e-6}
SAMPLES = (0, 7, 19, 33, 47)


@pytest.fixture(scope='module')
def backend():
    from sailfish_amd.backend_hip import HIPBackend

    class Opt(object):
        pass
    return HIPBackend(Opt(), 0)


def _one_step(backend, grid, size, f, precision, fused, **kw):
    """Every node = f, one AB step; returns the populations of the real nodes [Q, nodes]."""
    edm = kw.pop('edm', False)
    desc = make_box_desc(grid, size, precision=precision, access_pattern='AB', periodic_fused=[fused] * 3, **kw)
    if edm:
        desc.force_implementation = hipabi.SLF_FORCE_EDM
    s = BoxSim(backend, desc, periodic=(True, True, True))
    full = np.empty((s.Q,) + s.shape, dtype=s.dtype)
    full[...] = np.asarray(f, dtype=s.dtype).reshape((s.Q,) + (1,) * len(s.shape))
    s.set_dist(full, 0)
    s.set_dist(full, 1)
    s.step(save_macro=True)
    s.sync()
    out = s.real_view(s.get_dist()).reshape(s.Q, -1).astype(np.float64)
    rho, v = s.fetch_fields()
    macro = (float(s.real_view(rho).flat[0]), [float(s.real_view(c).flat[0]) for c in v])
    s.release()
    return out, macro


def _check(res, gold, tol, rho=None, v=None):
    out, (g_rho, g_v) = res
    if rho is not None:       # the macroscopic fields the same launch stores (pre-collision moments, u + a / 2)
        assert abs(g_rho - rho) < tol and max(abs(a - b) for a, b in zip(g_v, v)) < tol, (g_rho, rho, g_v, v)
    err = float(np.max(np.abs(out - np.asarray(gold, dtype=np.float64)[:, None])))
    assert err < tol, 'max err %.3e (tol %.1e)' % (err, tol)
    assert float(np.max(np.abs(out - out[:, :1]))) == 0.0          # every node did the same arithmetic


@pytest.mark.parametrize('fused', [1, 0], ids=['in_sweep_wrap', 'ghost_pbc'])
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_collision_kernels_against_reference_fixtures(backend, golden_dir, name, precision, fused):
    grid, size = GRIDS[name]
    G = np.load(os.path.join(golden_dir, 'arith_%s.npz' % name))
    tol = TOL[precision]
    mrt_tol = tol * (50 if precision == 'double' else 4)           # integer moment weights up to 30 (as for the oracle)
    for k in SAMPLES:
        f = G['f'][k]
        accel = list(G['accel'][k])
        for a, nu in enumerate(G['bgk_visc']):
            _check(_one_step(backend, grid, size, f, precision, fused, model='bgk', visc=float(nu)), G['bgk_post'][a, k], tol,
                   rho=float(G['mom_rho'][k]), v=list(G['mom_v'][k]))
            _check(_one_step(backend, grid, size, f, precision, fused, model='mrt', visc=float(nu)), G['mrt_post'][a, k],
                   mrt_tol)
        nu = float(G['guo_visc'][0])
        _check(_one_step(backend, grid, size, f, precision, fused, model='bgk', visc=nu, accel=accel), G['guo_post'][k], tol,
               rho=float(G['mom_rho'][k]), v=list(G['guo_out_v'][k]))
        _check(_one_step(backend, grid, size, f, precision, fused, model='bgk', visc=nu, accel=accel, edm=True),
               G['edm_post'][k], tol)
        nu = float(G['mrt_force_visc'][0])
        _check(_one_step(backend, grid, size, f, precision, fused, model='mrt', visc=nu, accel=accel),
               G['mrt_force_post'][k], mrt_tol)
