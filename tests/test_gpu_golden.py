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


def _one_step(backend, grid, size, f, precision, fused, box_cls=None, **kw):
    """Every node = f, one AB step; returns the populations of the real nodes [Q, nodes]."""
    edm = kw.pop('edm', False)
    desc = make_box_desc(grid, size, precision=precision, access_pattern='AB', periodic_fused=[fused] * 3, **kw)
    if edm:
        desc.force_implementation = hipabi.SLF_FORCE_EDM
    s = (box_cls or BoxSim)(backend, desc, periodic=(True, True, True))
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
    collision_probe(backend, golden_dir, name, precision, fused)


def collision_probe(backend, golden_dir, name, precision, fused, box_cls=None):
    grid, size = GRIDS[name]
    G = np.load(os.path.join(golden_dir, 'arith_%s.npz' % name))
    tol = TOL[precision]
    mrt_tol = tol * (50 if precision == 'double' else 4)           # integer moment weights up to 30 (as for the oracle)
    for k in SAMPLES:
        f = G['f'][k]
        accel = list(G['accel'][k])
        for a, nu in enumerate(G['bgk_visc']):
            _check(_one_step(backend, grid, size, f, precision, fused, box_cls, model='bgk', visc=float(nu)), G['bgk_post'][a, k], tol,
                   rho=float(G['mom_rho'][k]), v=list(G['mom_v'][k]))
            _check(_one_step(backend, grid, size, f, precision, fused, box_cls, model='mrt', visc=float(nu)), G['mrt_post'][a, k],
                   mrt_tol)
        nu = float(G['guo_visc'][0])
        _check(_one_step(backend, grid, size, f, precision, fused, box_cls, model='bgk', visc=nu, accel=accel), G['guo_post'][k], tol,
               rho=float(G['mom_rho'][k]), v=list(G['guo_out_v'][k]))
        _check(_one_step(backend, grid, size, f, precision, fused, box_cls, model='bgk', visc=nu, accel=accel, edm=True),
               G['edm_post'][k], tol)
        nu = float(G['mrt_force_visc'][0])
        _check(_one_step(backend, grid, size, f, precision, fused, box_cls, model='mrt', visc=nu, accel=accel),
               G['mrt_force_post'][k], mrt_tol)


@pytest.mark.parametrize('fused', [1, 0], ids=['in_sweep_wrap', 'ghost_pbc'])
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_minimize_roundoff_against_reference_fixtures(backend, golden_dir, name, precision, fused):
    roundoff_probe(backend, golden_dir, name, precision, fused)


def roundoff_probe(backend, golden_dir, name, precision, fused, box_cls=None):
    """--minimize_roundoff (reference lb_base.py:72-76): the arrays hold f_i - w_i.  Fixtures evaluated from the
    reference's sympy objects built with config.minimize_roundoff = True (tools/capture_goldens.py: ex_rho /
    ex_velocity of the shifted populations, bgk_equilibrium with rho0 = rho + 1, the Guo term with its prefactor):
    collision with and without a body force, the stored density delta / velocity, and SetInitialConditions."""
    grid, size = GRIDS[name]
    G = np.load(os.path.join(golden_dir, 'arith_%s.npz' % name))
    tol = TOL[precision]
    RO = hipabi.SLF_DENSITY_ROUNDOFF
    for k in SAMPLES:
        f = G['ro_f'][k]
        for a, nu in enumerate(G['ro_visc']):
            _check(_one_step(backend, grid, size, f, precision, fused, box_cls, model='bgk', visc=float(nu), incompressible=RO),
                   G['ro_bgk_post'][a, k], tol, rho=float(G['ro_mom_rho'][k]), v=list(G['ro_mom_v'][k]))
        _check(_one_step(backend, grid, size, f, precision, fused, box_cls, model='bgk', visc=float(G['ro_guo_visc'][0]),
                         accel=list(G['ro_accel'][k]), incompressible=RO),
               G['ro_guo_post'][k], tol, rho=float(G['ro_mom_rho'][k]), v=list(G['ro_guo_out_v'][k]))
        desc = make_box_desc(grid, size, precision=precision, access_pattern='AB', periodic_fused=[fused] * 3, incompressible=RO)
        s = (box_cls or BoxSim)(backend, desc, periodic=(True, True, True))
        shape = tuple(reversed(size))
        s.set_fields(np.full(shape, G['rho'][k]), [np.full(shape, G['v'][k][d]) for d in range(grid.dim)])
        s.initial_conditions()
        s.sync()
        out = s.real_view(s.get_dist()).reshape(s.Q, -1).astype(np.float64)
        assert float(np.max(np.abs(out - G['ro_feq'][k][:, None]))) < tol
        s.release()


BC_CASES = {    # kind: (type id in tests/_geometry.TYPE_KIND, parameter fixture, expected populations, expected rho, expected v)
    'regularized_velocity': ('T_REGVEL', 'bc_v', 'regvel_post', 'regvel_rho', 'bc_v'),
    'zouhe_velocity': ('T_ZHVEL', 'bc_v', 'zouhe_vel_post', 'regvel_rho', 'bc_v'),
    'equilibrium_density': ('T_EQDENS', 'bc_rho', 'eqdens_post', 'bc_rho', 'eqdens_v'),
    'zouhe_density': ('T_ZHDENS', 'bc_rho', 'zouhe_dens_post', 'bc_rho', 'zouhe_dens_v'),
    'regularized_density': ('T_REGDENS', 'bc_rho', 'regdens_post', 'bc_rho', 'eqdens_v'),
}


@pytest.mark.parametrize('kind', sorted(BC_CASES))
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_boundary_nodes_against_reference_fixtures(backend, golden_dir, name, precision, kind):
    boundary_probe(backend, golden_dir, name, precision, kind)


@pytest.mark.parametrize('model', ['bgk', 'mrt'])
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_composed_step_with_boundary_condition_and_relaxation(backend, golden_dir, name, precision, model):
    composed_probe(backend, golden_dir, name, precision, model)


def composed_probe(backend, golden_dir, name, precision, model, box_cls=None):
    """ONE node update with a boundary condition AND relaxation on, composed in the order of the reference's kernel
    (lb_single_fluid.mako:175-228: getMacro -> pre-collision boundary condition -> relaxate with the same rho / u ->
    stream): a regularized-velocity node (the lid of the cavity configurations) per orientation, BGK and MRT.  The
    single-piece fixtures pin moments, boundary conditions and collisions separately; this one pins how the kernel
    strings them together -- which density and velocity the collision sees after the boundary condition replaced the
    populations."""
    boundary_probe(backend, golden_dir, name, precision, 'regularized_velocity', box_cls=box_cls,
                   relaxation=model, fkey='regvel_%s_step' % model)


def boundary_probe(backend, golden_dir, name, precision, kind, box_cls=None, relaxation=None, fkey=None):
    """Single-node probes of the pre-collision boundary conditions (reference boundary.mako:343-382, 425-459, 784-878;
    fixtures from sym.noneq_bb / zouhe_fixup / reglb_flux_tensor / ex_rho): one boundary node per orientation in a
    box whose nodes all carry the fixture state f, relaxation switched off, one two-copy step through the node-map
    kernels; what the node pushed is collected from its neighbours (population i of x sits in slot i of x + e_i)."""
    from tests import _geometry as geo
    grid, _ = GRIDS[name]
    dim = grid.dim
    G = np.load(os.path.join(golden_dir, 'arith_%s.npz' % name))
    tname, pkey, fkey0, rkey, vkey = BC_CASES[kind]
    fkey = fkey or fkey0
    tol = TOL[precision] * (2 if relaxation != 'mrt' else (100 if precision == 'double' else 8))
    norient = 2 * dim
    size = (2 * norient + 3, 5) + ((5,) if dim == 3 else ())
    for k in SAMPLES[:3]:
        params = [float(x) for x in np.atleast_1d(G[pkey][k])]
        desc = make_box_desc(grid, size, precision=precision, access_pattern='AB', fluid_only=False,
                             type_kind=geo.TYPE_KIND, nt_bits=geo.NT_BITS, node_params=params,
                             relaxation_enabled=relaxation is not None, model=relaxation or 'bgk',
                             visc=float(G['step_visc'][0]))
        m = geo.empty_map(desc)
        probes = []
        for o in range(1, norient + 1):
            pos = (3 if dim == 3 else 0, 3, 2 * o)                       # (z, y, x) array indices, ghost layer included
            m[pos] = geo.encode(getattr(geo, tname), orientation=o, param=0)
            probes.append(pos)
        s = (box_cls or BoxSim)(backend, desc, periodic=(False, False, False), node_map=m)
        full = np.empty((s.Q,) + s.shape, dtype=s.dtype)
        full[...] = np.asarray(G['f'][k], dtype=s.dtype).reshape((s.Q,) + (1,) * len(s.shape))
        s.set_dist(full, 0)
        s.set_dist(full, 1)
        s.step(save_macro=True)
        out = s.get_dist().astype(np.float64)
        rho, v = s.fetch_fields()
        for o, (z, y, x) in enumerate(probes):
            got = np.array([out[i, z + (grid.basis[i][2] if dim == 3 else 0), y + grid.basis[i][1], x + grid.basis[i][0]]
                            for i in range(grid.Q)])
            err = float(np.max(np.abs(got - G[fkey][o, k])))
            assert err < tol, (kind, o + 1, k, err)
            want_rho = G[rkey][o, k] if G[rkey].ndim == 2 else G[rkey][k]
            want_v = G[vkey][o, k] if G[vkey].ndim == 3 else G[vkey][k]
            assert abs(float(rho[z, y, x]) - float(want_rho)) < max(tol, 1e-7)
            assert max(abs(float(v[d][z, y, x]) - float(want_v[d])) for d in range(dim)) < max(tol, 1e-7)
        s.release()


RO_TYPE_KIND = [hipabi.SLF_NK_FLUID, hipabi.SLF_NK_GHOST, hipabi.SLF_NK_FULL_BB, hipabi.SLF_NK_EQUILIBRIUM_VELOCITY,
                hipabi.SLF_NK_EQUILIBRIUM_DENSITY]
RO_BC_CASES = {   # kind: (type id in RO_TYPE_KIND, parameter fixture, expected populations, expected density delta, expected v)
    'equilibrium_velocity': (3, 'bc_v', 'eqvel_post', 'eqvel_drho', 'bc_v'),
    'equilibrium_density': (4, 'bc_rho', 'eqdens_post', 'bc_drho', 'eqdens_v'),
}


@pytest.mark.parametrize('kind', sorted(RO_BC_CASES))
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_minimize_roundoff_boundary_nodes_against_reference_fixtures(backend, golden_dir, name, precision, kind):
    roundoff_boundary_probe(backend, golden_dir, name, precision, kind)


def roundoff_boundary_probe(backend, golden_dir, name, precision, kind, box_cls=None):
    """Equilibrium velocity / density nodes under --minimize_roundoff (boundary.mako:420-506, 797-809 with
    config.minimize_roundoff; fixtures tools/capture_goldens.py: roundoff_bc_goldens): single-node probes as
    boundary_probe(), the arrays hold f_i - w_i, the stored density is the delta."""
    from tests import _geometry as geo
    grid, _ = GRIDS[name]
    dim = grid.dim
    G = np.load(os.path.join(golden_dir, 'arith_ro_bc_%s.npz' % name))
    tid, pkey, fkey, rkey, vkey = RO_BC_CASES[kind]
    tol = TOL[precision] * 2
    norient = 2 * dim
    size = (2 * norient + 3, 5) + ((5,) if dim == 3 else ())
    for k in range(3):
        params = [float(x) for x in np.atleast_1d(G[pkey][k])]
        desc = make_box_desc(grid, size, precision=precision, access_pattern='AB', fluid_only=False, type_kind=RO_TYPE_KIND,
                             nt_bits=geo.NT_BITS, node_params=params, relaxation_enabled=False, model='bgk',
                             incompressible=hipabi.SLF_DENSITY_ROUNDOFF)
        m = geo.empty_map(desc)
        probes = []
        for o in range(1, norient + 1):
            pos = (3 if dim == 3 else 0, 3, 2 * o)
            m[pos] = geo.encode(tid, orientation=o, param=0)
            probes.append(pos)
        s = (box_cls or BoxSim)(backend, desc, periodic=(False, False, False), node_map=m)
        full = np.empty((s.Q,) + s.shape, dtype=s.dtype)
        full[...] = np.asarray(G['f'][k], dtype=s.dtype).reshape((s.Q,) + (1,) * len(s.shape))
        s.set_dist(full, 0)
        s.set_dist(full, 1)
        s.step(save_macro=True)
        out = s.get_dist().astype(np.float64)
        rho, v = s.fetch_fields()
        for o, (z, y, x) in enumerate(probes):
            got = np.array([out[i, z + (grid.basis[i][2] if dim == 3 else 0), y + grid.basis[i][1], x + grid.basis[i][0]]
                            for i in range(grid.Q)])
            err = float(np.max(np.abs(got - G[fkey][o, k])))
            assert err < tol, (kind, o + 1, k, err)
            want_rho = G[rkey][o, k] if G[rkey].ndim == 2 else G[rkey][k]
            want_v = G[vkey][o, k] if G[vkey].ndim == 3 else G[vkey][k]
            assert abs(float(rho[z, y, x]) - float(want_rho)) < max(tol, 1e-7)
            assert max(abs(float(v[d][z, y, x]) - float(want_v[d])) for d in range(dim)) < max(tol, 1e-7)
        s.release()


@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_initial_conditions_against_reference_fixtures(backend, golden_dir, name, precision):
    init_probe(backend, golden_dir, name, precision)


def init_probe(backend, golden_dir, name, precision, box_cls=None):
    """SetInitialConditions (reference lb_single_fluid.mako:101-127): f = feq(rho, v) from sym_equilibrium.bgk_equilibrium,
    compressible and incompressible."""
    grid, size = GRIDS[name]
    G = np.load(os.path.join(golden_dir, 'arith_%s.npz' % name))
    for inc in (0, 1):
        for k in SAMPLES:
            desc = make_box_desc(grid, size, precision=precision, access_pattern='AB', periodic_fused=[1] * 3,
                                 incompressible=bool(inc))
            s = (box_cls or BoxSim)(backend, desc, periodic=(True, True, True))
            shape = tuple(reversed(size))
            s.set_fields(np.full(shape, G['rho'][k]), [np.full(shape, G['v'][k][d]) for d in range(grid.dim)])
            s.initial_conditions()
            s.sync()
            out = s.real_view(s.get_dist()).reshape(s.Q, -1).astype(np.float64)
            err = float(np.max(np.abs(out - G['feq_inc%d' % inc][k][:, None])))
            assert err < TOL[precision], (inc, k, err)
            s.release()
