#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json from the reference's importable symbolic /
host layer (sympy expressions, node-map encoder, connection tables).

Runs ONLY in the authoring container (needs /root/reference); the produced
fixtures are data (inputs + expected outputs) and are committed.  Nothing in
tests/ or the product reads /root/reference at run time.

Every expected value is produced by *evaluating the reference's own sympy
expression objects* (sailfish/sym.py, sym_equilibrium.py, sym_force.py) in
float64; the composition order follows the reference's templates (cited inline).

    PYTHONPATH=tools python tools/capture_goldens.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: F401  (installs import stubs, puts /root/reference on sys.path)

import numpy as np
import sympy

from sailfish import sym, sym_equilibrium, sym_force  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
S = sym.S


class _Cfg(object):
    def __init__(self, incompressible=False, minimize_roundoff=False):
        self.incompressible = incompressible
        self.minimize_roundoff = minimize_roundoff


def _evalf(expr, subs):
    """Evaluate a sympy expression in float64 after substituting symbols by name."""
    if isinstance(expr, (int, float)):
        return float(expr)
    e = expr
    m = {}
    for s in e.free_symbols:
        if s.name not in subs:
            raise KeyError('unbound symbol %s in %s' % (s.name, expr))
        m[s] = sympy.Float(subs[s.name], 30)
    return float(e.subs(m).evalf(30))


def _fi_subs(grid, f, ptr='fi'):
    return {'%s->%s' % (ptr, n): float(v) for n, v in zip(grid.idx_name, f)}


def _macro_subs(grid, rho, v, rho0=None):
    d = {'g0m0': float(rho), 'rho': float(rho)}
    for c, val in zip('xyz', v):
        d['g0m1' + c] = float(val)
    d['rho0'] = float(rho if rho0 is None else rho0)
    return d


def lattice_tables(grid):
    """Lattice algebra pinned by reference tests/sym.py:6-67."""
    dim = grid.dim
    t = {
        'name': grid.__name__, 'dim': dim, 'Q': grid.Q,
        'basis': [[int(c) for c in e] for e in grid.basis],
        'weights_num': [int(w.p) for w in grid.weights],
        'weights_den': [int(w.q) for w in grid.weights],
        'idx_name': list(grid.idx_name),
        'idx_opposite': [int(i) for i in grid.idx_opposite],
        'dir2vecidx': {str(k): int(v) for k, v in grid.dir2vecidx.items()},
        'dir_to_vec': {str(o): [int(c) for c in grid.dir_to_vec(o)] for o in range(1, 2 * dim + 1)},
        'bb_swap_pairs': sorted(int(i) for i in sym.bb_swap_pairs(grid)),
        'prop_dists': {('%d_%d' % (axis, d)): [int(i) for i in sym.get_prop_dists(grid, d, axis)]
                       for axis in range(dim) for d in (-1, 0, 1)},
        'missing_dists': {str(o): [int(i) for i in sym.get_missing_dists(grid, o)]
                          for o in range(1, 2 * dim + 1)},
        # full-slip node: which populations swap, per orientation (boundary.mako:837-855)
        'slip_swap_pairs': {str(o): sorted([int(i), int(j)] for i, j in sym.slip_bb_swap_pairs(grid, o))
                            for o in range(1, 2 * dim + 1)},
    }
    import itertools
    ib = {}
    for d in itertools.product((-1, 0, 1), repeat=dim):
        if not any(d):
            continue
        key = ','.join(str(c) for c in d)
        ib[key] = {'normal': [int(i) for i in sym.get_interblock_dists(grid, d)],
                   'opposite': [int(i) for i in sym.get_interblock_dists(grid, d, opposite=True)]}
    t['interblock_dists'] = ib
    if hasattr(grid, 'mrt_matrix'):
        t['mrt_names'] = list(grid.mrt_names)
        t['mrt_matrix'] = [[int(x) for x in grid.mrt_matrix.row(i)] for i in range(grid.Q)]
        coll = []
        for c in grid.mrt_collision:
            coll.append('inv_tau' if isinstance(c, sympy.Basic) and c.free_symbols else float(c))
        t['mrt_collision'] = coll
    return t


def arithmetic_goldens(grid, rng, n=48):
    dim, Q = grid.dim, grid.Q
    out = {}
    rho = rng.uniform(0.9, 1.1, n)
    v = rng.uniform(-0.1, 0.1, (n, dim))
    f = np.array([[float(w) for w in grid.weights]] * n) * rho[:, None]
    f = f * (1.0 + rng.uniform(-0.05, 0.05, (n, Q)))
    out['rho'], out['v'], out['f'] = rho, v, f

    # --- equilibrium: sym_equilibrium.py:90-120, compressible and incompressible
    for inc in (False, True):
        cfg = _Cfg(incompressible=inc)
        eq = sym_equilibrium.bgk_equilibrium(grid, cfg)
        res = np.zeros((n, Q))
        for k in range(n):
            subs = _macro_subs(grid, rho[k], v[k], rho0=(1.0 if inc else rho[k]))
            for i, e in enumerate(eq.expression):
                res[k, i] = _evalf(e, subs)
        out['feq_inc%d' % int(inc)] = res

    # --- the --minimize_roundoff formulation (reference lb_base.py:72-76): every expression built with
    #     config.minimize_roundoff = True -- equilibrium at (delta rho, v) (sym_equilibrium.py:100-118, rho0 = rho + 1),
    #     moments of the shifted populations f - w (sym.py:573-597, 654-661), BGK collision, Guo forcing with the
    #     prefactor of sym_force.py:147-160
    cfg_ro = _Cfg(minimize_roundoff=True)
    wts = np.array([float(w) for w in grid.weights])
    f_ro = f - wts[None, :]
    eq_ro = sym_equilibrium.bgk_equilibrium(grid, cfg_ro)
    ro_rho = np.zeros(n)
    ro_v = np.zeros((n, dim))
    ro_feq = np.zeros((n, Q))
    for k in range(n):
        subs = _fi_subs(grid, f_ro[k])
        ro_rho[k] = _evalf(sym.ex_rho(grid, 'fi', False, minimize_roundoff=True), subs)
        subs2 = dict(subs)
        subs2.update({'g0m0': ro_rho[k], 'rho': ro_rho[k]})
        for d in range(dim):
            ro_v[k, d] = _evalf(sym.ex_velocity(grid, 'fi', d, cfg_ro), subs2)
        subs = _macro_subs(grid, rho[k] - 1.0, v[k])
        for i, e in enumerate(eq_ro.expression):
            ro_feq[k, i] = _evalf(e, subs)
    out['ro_f'], out['ro_mom_rho'], out['ro_mom_v'], out['ro_feq'] = f_ro, ro_rho, ro_v, ro_feq
    ro_visc = np.array([1.0 / 6.0, 0.01])
    ro_post = np.zeros((len(ro_visc), n, Q))
    for a, nu in enumerate(ro_visc):
        omega = 1.0 / sym.relaxation_time(nu)
        for k in range(n):
            subs = _macro_subs(grid, ro_rho[k], ro_v[k])
            for i, e in enumerate(eq_ro.expression):
                ro_post[a, k, i] = f_ro[k, i] + omega * (_evalf(e, subs) - f_ro[k, i])
    out['ro_visc'], out['ro_bgk_post'] = ro_visc, ro_post
    ro_accel = np.random.RandomState(77).uniform(-1e-4, 1e-4, (n, dim))
    ro_guo = np.zeros((n, Q))
    nu = 0.02
    tau = sym.relaxation_time(nu)
    guo_e = sym_force.guo_external_force(grid, grid_num=0)
    pref_ro = sym_force.guo_external_force_pref(grid, cfg_ro, grid_num=0)
    for k in range(n):
        v0 = ro_v[k] + 0.5 * ro_accel[k]
        subs = _macro_subs(grid, ro_rho[k], v0)
        subs.update({'g0ea' + c: ro_accel[k, j] for j, c in enumerate('xyz'[:dim])})
        subs['tau0'] = tau
        subs['pref'] = _evalf(pref_ro, subs)
        for i, e in enumerate(eq_ro.expression):
            ro_guo[k, i] = f_ro[k, i] + (1.0 / tau) * (_evalf(e, subs) - f_ro[k, i]) + _evalf(guo_e[i], subs)
    out['ro_accel'], out['ro_guo_post'], out['ro_guo_out_v'] = ro_accel, ro_guo, ro_v + 0.5 * ro_accel
    out['ro_guo_visc'] = np.array([nu])

    cfg = _Cfg()
    # --- moments: sym.py:573-682
    ex_rho = sym.ex_rho(grid, 'fi', False)
    ex_v = [sym.ex_velocity(grid, 'fi', d, cfg) for d in range(dim)]
    ex_mom = [sym.ex_velocity(grid, 'fi', d, cfg, momentum=True) for d in range(dim)]
    m_rho = np.zeros(n)
    m_v = np.zeros((n, dim))
    m_mom = np.zeros((n, dim))
    for k in range(n):
        subs = _fi_subs(grid, f[k])
        m_rho[k] = _evalf(ex_rho, subs)
        subs2 = dict(subs)
        subs2.update({'g0m0': m_rho[k], 'rho': m_rho[k]})
        for d in range(dim):
            m_v[k, d] = _evalf(ex_v[d], subs2)
            m_mom[k, d] = _evalf(ex_mom[d], subs2)
    out['mom_rho'], out['mom_v'], out['mom_momentum'] = m_rho, m_v, m_mom

    # --- 2nd moments: sym.py:684-704
    npairs = dim * (dim + 1) // 2
    flux = np.zeros((n, npairs))
    eqflux = np.zeros((n, npairs))
    for k in range(n):
        subs = _fi_subs(grid, f[k])
        ms = _macro_subs(grid, rho[k], v[k])
        j = 0
        for a in range(dim):
            for b in range(a, dim):
                flux[k, j] = _evalf(sym.ex_flux(grid, 'fi', a, b, cfg), subs)
                eqflux[k, j] = _evalf(sym.ex_eq_flux(grid, a, b), ms)
                j += 1
    out['flux'], out['eq_flux'] = flux, eqflux

    # --- BGK collision: relaxation.mako:127-132, tau = (6 visc + 1)/2 (sym.py:847-848)
    visc = np.array([1.0 / 6.0, 0.01, 0.0254, 0.1])
    out['bgk_visc'] = visc
    post = np.zeros((len(visc), n, Q))
    eq = sym_equilibrium.bgk_equilibrium(grid, cfg)
    for a, nu in enumerate(visc):
        tau = sym.relaxation_time(nu)
        omega = 1.0 / tau
        for k in range(n):
            subs = _macro_subs(grid, m_rho[k], m_v[k])
            for i, e in enumerate(eq.expression):
                feq = _evalf(e, subs)
                post[a, k, i] = f[k, i] + omega * (feq - f[k, i])
    out['bgk_post'] = post

    # --- BGK + Guo body force: relaxation_common.mako:56-64,110-149, sym_force.py:121-160
    accel = rng.uniform(-1e-4, 1e-4, (n, dim))
    out['accel'] = accel
    postf = np.zeros((n, Q))
    outv = np.zeros((n, dim))
    nu = 0.02
    tau = sym.relaxation_time(nu)
    out['guo_visc'] = np.array([nu])
    guo = sym_force.guo_external_force(grid, grid_num=0)
    pref_e = sym_force.guo_external_force_pref(grid, cfg, grid_num=0)
    for k in range(n):
        v0 = m_v[k] + 0.5 * accel[k]
        subs = _macro_subs(grid, m_rho[k], v0)
        subs.update({'g0ea' + c: accel[k, j] for j, c in enumerate('xyz'[:dim])})
        subs['tau0'] = tau
        pref = _evalf(pref_e, subs)
        subs['pref'] = pref
        for i, e in enumerate(eq.expression):
            feq = _evalf(e, subs)
            postf[k, i] = f[k, i] + (1.0 / tau) * (feq - f[k, i]) + _evalf(guo[i], subs)
        outv[k] = m_v[k] + 0.5 * accel[k]
    out['guo_post'], out['guo_out_v'] = postf, outv

    # --- BGK + exact difference method: relaxation_common.mako:66-73,95-99, sym_force.py:184-193
    #     (equilibrium at the unshifted velocity, f_i += feq_i(u + a) - feq_i(u))
    edm = sym_force.edm_shift_velocity(eq.expression, grid, 0)
    poste = np.zeros((n, Q))
    for k in range(n):
        subs = _macro_subs(grid, m_rho[k], m_v[k])
        subs.update({'g0ea' + c: accel[k, j] for j, c in enumerate('xyz'[:dim])})
        for i, e in enumerate(eq.expression):
            feq = _evalf(e, subs)
            poste[k, i] = f[k, i] + (1.0 / tau) * (feq - f[k, i]) + (_evalf(edm[i], subs) - feq)
    out['edm_post'] = poste

    # --- MRT: relaxation_mrt.mako:31-97, sym.py:716-735 + grid.mrt_*
    if hasattr(grid, 'mrt_matrix'):
        M = np.array(grid.mrt_matrix.tolist(), dtype=np.float64)
        Minv = np.array(grid.mrt_matrix.inv().tolist(), dtype=np.float64)
        out['mrt_matrix'] = M
        out['mrt_matrix_inv'] = Minv
        postm = np.zeros((len(visc), n, Q))
        for a, nu in enumerate(visc):
            for k in range(n):
                m = M.dot(f[k])
                subs = {'mx': m[grid.mrt_names.index('mx')], 'my': m[grid.mrt_names.index('my')],
                        'rho': m[0], 'g0m0': m[0], 'rho0': m[0], 'visc': nu}
                if dim == 3:
                    subs['mz'] = m[grid.mrt_names.index('mz')]
                for lv in grid.mrt_eq_symbols:
                    subs[lv.lhs.name] = _evalf(lv.rhs, subs)
                for i in range(Q):
                    c = grid.mrt_collision[i]
                    cval = _evalf(c, subs) if isinstance(c, sympy.Basic) else float(c)
                    if cval != 0:
                        meq = _evalf(grid.mrt_equilibrium[i], subs)
                        m[i] -= cval * (m[i] - meq)
                postm[a, k] = Minv.dot(m)
        out['mrt_post'] = postm
        # with a body force (relaxation_mrt.mako:10-27 fluid_momentum(), :45-46, :91): half of the acceleration
        # sym_force.accel_vector(grid, 0) goes into the momentum moments before the equilibrium is evaluated, the
        # other half after the relaxation; the output velocity is shifted by a / 2 as for BGK
        ea = sym_force.accel_vector(grid, 0)
        mom = [grid.mrt_names.index('m' + c) for c in 'xyz'[:dim]]
        nu = visc[len(visc) // 2]
        postf = np.zeros((n, Q))
        for k in range(n):
            asub = {'g0ea' + c: accel[k, j] for j, c in enumerate('xyz'[:dim])}
            half = [0.5 * _evalf(ea[j], asub) for j in range(dim)]
            m = M.dot(f[k])
            for j in range(dim):
                m[mom[j]] += half[j]
            subs = {'rho': m[0], 'g0m0': m[0], 'rho0': m[0], 'visc': nu}
            subs.update({'m' + c: m[mom[j]] for j, c in enumerate('xyz'[:dim])})
            for lv in grid.mrt_eq_symbols:
                subs[lv.lhs.name] = _evalf(lv.rhs, subs)
            for i in range(Q):
                c = grid.mrt_collision[i]
                cval = _evalf(c, subs) if isinstance(c, sympy.Basic) else float(c)
                if cval != 0:
                    m[i] -= cval * (m[i] - _evalf(grid.mrt_equilibrium[i], subs))
            for j in range(dim):
                m[mom[j]] += half[j]
            postf[k] = Minv.dot(m)
        out['mrt_force_post'] = postf
        out['mrt_force_visc'] = np.array([nu])

    # --- velocity-BC density: boundary.mako:443-459, sym.py:621-627
    # --- non-equilibrium bounce-back: sym.py:750-766; regularisation: sym.py:882-891,
    #     boundary.mako:817-835  (NTRegularizedVelocity full node update, pre-collision)
    # --- density BC: boundary.mako:425-441, 495-500, 797-809 (NTEquilibriumDensity)
    eqx = eq.expression
    reg = sym.reglb_flux_tensor(grid)
    regv = np.zeros((2 * dim, n, Q))
    regv_rho = np.zeros((2 * dim, n))
    eqd = np.zeros((2 * dim, n, Q))
    eqd_v = np.zeros((2 * dim, n, dim))
    # --- Zou-He nodes: boundary.mako:343-382 (zouhe_bb), :487-494 (density variant), sym.py:768-815
    # --- regularized density: boundary.mako:501-506 + the regularisation above
    zhv = np.zeros((2 * dim, n, Q))
    zhd = np.zeros((2 * dim, n, Q))
    zhd_v = np.zeros((2 * dim, n, dim))
    regd = np.zeros((2 * dim, n, Q))

    def zouhe(o, g, rho_full, v, subs):
        """zouhe_bb after the non-equilibrium bounce-back: momentum difference + fix-up."""
        subs = dict(subs)
        subs.update(_fi_subs(grid, g))
        mom = [_evalf(sym.ex_velocity(grid, 'fi', d, cfg, momentum=True), subs) for d in range(dim)]
        for d, name in enumerate(['nvx', 'nvy', 'nvz'][:dim]):
            subs[name] = rho_full * v[d] - mom[d]
        g = g.copy()
        for lhs, rhs in sym.zouhe_fixup(grid, o):
            val = _evalf(rhs, subs)
            if lhs.name.startswith('fi->'):
                g[grid.idx_name.index(lhs.name.split('->')[1])] = val
            subs[lhs.name] = val
        return g
    bc_v = rng.uniform(-0.08, 0.08, (n, dim))
    bc_rho = rng.uniform(0.95, 1.05, n)
    out['bc_v'], out['bc_rho'] = bc_v, bc_rho
    for o in range(1, 2 * dim + 1):
        missing = sym.get_missing_dists(grid, o)
        nvec = [int(c) for c in grid.dir_to_vec(o)]
        for k in range(n):
            fi = f[k].copy()
            for i in missing:
                fi[i] = fi[grid.idx_opposite[i]]
            rs = _evalf(ex_rho, _fi_subs(grid, fi))
            # velocity BC
            subs = _macro_subs(grid, rs, bc_v[k])
            r = _evalf(sym.ex_rho(grid, 'fi', False, missing_dir=o), subs)
            regv_rho[o - 1, k] = r
            subs = _macro_subs(grid, r, bc_v[k])
            subs.update(_fi_subs(grid, fi))
            g = fi.copy()
            for lhs, rhs in sym.noneq_bb(grid, o, eqx):
                i = grid.idx_name.index(lhs.name.split('->')[1])
                g[i] = _evalf(rhs, subs)
            subs.update(_fi_subs(grid, g))
            fl = []
            for a in range(dim):
                for b in range(a, dim):
                    fl.append(_evalf(sym.ex_flux(grid, 'fi', a, b, cfg), subs)
                              - _evalf(sym.ex_eq_flux(grid, a, b), subs))
            subs.update({'flux[%d]' % j: x for j, x in enumerate(fl)})
            for i in range(Q):
                regv[o - 1, k, i] = max(1e-7, _evalf(eqx[i], subs) + _evalf(reg[i], subs))
            # Zou-He velocity node: g holds the populations after noneq_bb with (r, bc_v)
            subs_v = _macro_subs(grid, r, bc_v[k])
            subs_v.update(_fi_subs(grid, fi))
            g_v = fi.copy()
            for lhs, rhs in sym.noneq_bb(grid, o, eqx):
                g_v[grid.idx_name.index(lhs.name.split('->')[1])] = _evalf(rhs, subs_v)
            zhv[o - 1, k] = zouhe(o, g_v, r, bc_v[k], subs_v)
            # density BC
            subs = {'g0m0': rs, 'rho': rs, 'par_rho': bc_rho[k]}
            vv = [_evalf(sym.ex_velocity(grid, 'fi', d, cfg, missing_dir=o, par_rho='par_rho'), subs)
                  for d in range(dim)]
            eqd_v[o - 1, k] = vv
            subs = _macro_subs(grid, bc_rho[k], vv)
            for i in range(Q):
                eqd[o - 1, k, i] = _evalf(eqx[i], subs)
            # Zou-He density node: noneq_bb with (par_rho, vv), fix-up, then the standard moments
            subs_d = _macro_subs(grid, bc_rho[k], vv)
            subs_d.update(_fi_subs(grid, fi))
            g_d = fi.copy()
            for lhs, rhs in sym.noneq_bb(grid, o, eqx):
                g_d[grid.idx_name.index(lhs.name.split('->')[1])] = _evalf(rhs, subs_d)
            g_z = zouhe(o, g_d, bc_rho[k], vv, subs_d)
            zhd[o - 1, k] = g_z
            rz = _evalf(ex_rho, _fi_subs(grid, g_z))
            sz = _fi_subs(grid, g_z)
            sz.update({'rho': rz, 'g0m0': rz, 'rho0': rz})
            zhd_v[o - 1, k] = [_evalf(sym.ex_velocity(grid, 'fi', d, cfg), sz) for d in range(dim)]
            # regularized density node: regularisation of g_d with (par_rho, vv)
            subs_r = _macro_subs(grid, bc_rho[k], vv)
            subs_r.update(_fi_subs(grid, g_d))
            fl = []
            for a in range(dim):
                for b in range(a, dim):
                    fl.append(_evalf(sym.ex_flux(grid, 'fi', a, b, cfg), subs_r)
                              - _evalf(sym.ex_eq_flux(grid, a, b), subs_r))
            subs_r.update({'flux[%d]' % j: x for j, x in enumerate(fl)})
            for i in range(Q):
                regd[o - 1, k, i] = max(1e-7, _evalf(eqx[i], subs_r) + _evalf(reg[i], subs_r))
            del nvec[:]
    out['regvel_rho'], out['regvel_post'] = regv_rho, regv
    # --- one COMPOSED node update with a boundary condition and relaxation on, in the order of
    #     lb_single_fluid.mako:175-228: getMacro (rho from ex_rho(missing_dir), v = the node's parameters) ->
    #     precollisionBoundaryConditions (noneq_bb + regularisation: regv above) -> relaxate with the SAME g0m0 / v
    #     (BGK: relaxation.mako:127-132; MRT: moments of the regularised populations, relaxation_mrt.mako:31-97)
    nu_step = 0.02
    omega = 1.0 / sym.relaxation_time(nu_step)
    step_bgk = np.zeros((2 * dim, n, Q))
    step_mrt = np.zeros((2 * dim, n, Q))
    for o in range(1, 2 * dim + 1):
        for k in range(n):
            g = regv[o - 1, k]
            subs = _macro_subs(grid, regv_rho[o - 1, k], bc_v[k])
            for i in range(Q):
                step_bgk[o - 1, k, i] = g[i] + omega * (_evalf(eqx[i], subs) - g[i])
            if hasattr(grid, 'mrt_matrix'):
                m = M.dot(g)
                subs = {'rho': m[0], 'g0m0': m[0], 'rho0': m[0], 'visc': nu_step}
                subs.update({'m' + c: m[grid.mrt_names.index('m' + c)] for c in 'xyz'[:dim]})
                for lv in grid.mrt_eq_symbols:
                    subs[lv.lhs.name] = _evalf(lv.rhs, subs)
                for i in range(Q):
                    c = grid.mrt_collision[i]
                    cval = _evalf(c, subs) if isinstance(c, sympy.Basic) else float(c)
                    if cval != 0:
                        m[i] -= cval * (m[i] - _evalf(grid.mrt_equilibrium[i], subs))
                step_mrt[o - 1, k] = Minv.dot(m)
    out['regvel_bgk_step'], out['regvel_mrt_step'], out['step_visc'] = step_bgk, step_mrt, np.array([nu_step])
    out['eqdens_v'], out['eqdens_post'] = eqd_v, eqd
    out['zouhe_vel_post'], out['zouhe_dens_post'], out['zouhe_dens_v'] = zhv, zhd, zhd_v
    out['regdens_post'] = regd
    return out


def shan_chen_goldens(grid, rng, n=24):
    """Binary Shan-Chen node update composed from the reference's own objects: pseudopotentials
    sym.SHAN_CHEN_POTENTIALS (sym.py:896-908), moments sym.ex_rho / ex_velocity, the second lattice's equilibrium
    bgk_equilibrium(rho=S.phi, rho0=S.phi) (lb_binary.py:378-380), the per-lattice Guo term
    sym_force.guo_external_force(grid, grid_num=k) with guo_external_force_pref, in the order of
    binary_shan_chen.mako:69-83 (common velocity), shan_chen.mako:9-84 (force -> acceleration) and
    relaxation_common.mako:110-149 (equilibrium at v + a/2, Guo source term)."""
    from functools import partial
    dim, Q = grid.dim, grid.Q
    cfg = _Cfg()
    out = {}
    w = np.array([float(x) for x in grid.weights])
    f = [w[None, :] * rng.uniform(0.8, 1.2, (n, Q)) * rng.uniform(0.7, 1.3, (n, 1)) for _ in range(2)]
    nb = [rng.uniform(0.4, 1.6, (n, Q)) for _ in range(2)]        # rho / phi at x + e_i (index 0 unused)
    G = rng.uniform(-1.0, 1.5, (n, 4))
    G[:, 2] = G[:, 1]                                                # G21 = G12 (lb_binary.py:393)
    G[n // 2:, 0] = 0.0                                              # couplings that are switched off are skipped
    G[n // 3:2 * n // 3, 3] = 0.0
    body = rng.uniform(-2e-5, 2e-5, (n, 2, dim))
    body[:n // 2] = 0.0
    visc, tau_phi = 0.07, 0.83
    tau = [float(sym.relaxation_time(visc)), tau_phi]
    out.update(f1=f[0], f2=f[1], rho_nb=nb[0], phi_nb=nb[1], G=G, body_accel=body, visc=np.array([visc]),
               tau_phi=np.array([tau_phi]))
    eqs = [sym_equilibrium.bgk_equilibrium(grid, cfg),
           partial(sym_equilibrium.bgk_equilibrium, rho=S.phi, rho0=S.phi)(grid, cfg)]
    guo = [sym_force.guo_external_force(grid, grid_num=k) for k in range(2)]
    prefs = [sym_force.guo_external_force_pref(grid, cfg, grid_num=k) for k in range(2)]
    rho_e = sym.ex_rho(grid, 'fi', False)
    # macroscopic pass
    rho = np.zeros((n, 2))
    mom = np.zeros((n, 2, dim))
    for k in range(n):
        for l in range(2):
            subs = _fi_subs(grid, f[l][k])
            rho[k, l] = _evalf(rho_e, subs)
            for d in range(dim):
                mom[k, l, d] = _evalf(sym.ex_velocity(grid, 'fi', d, cfg, momentum=True), subs)
    ti = [1.0 / tau[0], 1.0 / tau[1]]
    vc = (ti[0] * mom[:, 0] + ti[1] * mom[:, 1]) / (ti[0] * rho[:, 0] + ti[1] * rho[:, 1])[:, None]
    out.update(sc_rho=rho[:, 0], sc_phi=rho[:, 1], sc_v=vc)
    basis = [np.array([int(c) for c in e]) for e in grid.basis]
    for pot in ('linear', 'classic'):
        pe = sym.SHAN_CHEN_POTENTIALS[pot]('lfield')
        psi = lambda x: _evalf(pe, {'lfield': x})     # noqa: E731
        acc = np.zeros((n, 2, dim))
        post = np.zeros((n, 2, Q))
        for k in range(n):
            for l in range(2):
                a = np.zeros(dim)
                for j in range(2):
                    cc = G[k, 2 * l + j]
                    if cc == 0.0:
                        continue
                    force = np.zeros(dim)
                    for i in range(1, Q):
                        p = psi(nb[j][k, i])
                        for d in range(dim):
                            if basis[i][d] != 0:
                                force[d] += p * float(basis[i][d] * grid.weights[i])
                    a += force * (-psi(rho[k, l]) * cc)
                a = a / rho[k, l] + body[k, l]
                acc[k, l] = a
                v0 = vc[k] + 0.5 * a
                subs = {'g0m0': rho[k, 0], 'g1m0': rho[k, 1], 'rho': rho[k, l], 'rho0': rho[k, l],
                        'tau0': tau[0], 'tau1': tau[1]}
                for d, c in enumerate('xyz'[:dim]):
                    subs['g0m1' + c] = v0[d]
                    subs['g%dea%s' % (l, c)] = a[d]
                subs['pref'] = _evalf(prefs[l], subs)
                for i in range(Q):
                    feq = _evalf(eqs[l].expression[i], subs)
                    post[k, l, i] = f[l][k, i] + ti[l] * (feq - f[l][k, i]) + _evalf(guo[l][i], subs)
        out['sc_accel_' + pot] = acc
        out['sc_post_' + pot] = post
    xs = rng.uniform(0.1, 2.0, n)
    out['psi_in'] = xs
    for pot in ('linear', 'classic'):
        pe = sym.SHAN_CHEN_POTENTIALS[pot]('lfield')
        out['psi_' + pot] = np.array([_evalf(pe, {'lfield': x}) for x in xs])
    return out


def roundoff_bc_goldens(grid, rng, n=12):
    """Boundary-condition nodes under --minimize_roundoff (config.minimize_roundoff = True; the populations are
    f_i - w_i, the density variable is rho - 1), evaluated from the reference's expression objects in the order of
    boundary.mako:420-506 (getMacro) and :797-809 (equilibrium nodes):
      equilibrium VELOCITY node: fill the unknown populations with their opposites; rho = ex_rho(missing_dir,
        minimize_roundoff) = (sum + n.v) / (1 - n.v); f := bgk_equilibrium(cfg)(rho, v_bc)
      equilibrium DENSITY node: v from ex_velocity(missing_dir, par_rho) (= -n (sum + 1 - par_rho) / par_rho);
        rho = par_rho - 1 (boundary.mako:492-504); f := equilibrium
    and, as evidence for what is NOT offered, the reference's REGULARIZED velocity node under the option applied to a
    node that is exactly at equilibrium: the standard formulation returns the equilibrium (the non-equilibrium flux
    vanishes); with minimize_roundoff ex_flux adds c_s^2 (sym.py:684-695) while ex_eq_flux keeps multiplying by the
    density DELTA (sym.py:697-703), so the "non-equilibrium" flux is off by O(1) and the node comes out wrong."""
    dim, Q = grid.dim, grid.Q
    cfg_ro, cfg = _Cfg(minimize_roundoff=True), _Cfg()
    eq_ro = sym_equilibrium.bgk_equilibrium(grid, cfg_ro).expression
    eq_std = sym_equilibrium.bgk_equilibrium(grid, cfg).expression
    wts = np.array([float(w) for w in grid.weights])
    rho = rng.uniform(0.9, 1.1, n)
    f = wts[None, :] * rho[:, None] * (1.0 + rng.uniform(-0.05, 0.05, (n, Q)))
    bc_v = rng.uniform(-0.08, 0.08, (n, dim))
    bc_rho = rng.uniform(0.95, 1.05, n)
    out = {'f': f - wts[None, :], 'bc_v': bc_v, 'bc_rho': bc_rho, 'bc_drho': bc_rho - 1.0}
    eqv_rho = np.zeros((2 * dim, n))
    eqv = np.zeros((2 * dim, n, Q))
    eqd_v = np.zeros((2 * dim, n, dim))
    eqd = np.zeros((2 * dim, n, Q))
    reg_defect = np.zeros((2 * dim, 2))
    reg = sym.reglb_flux_tensor(grid)
    for o in range(1, 2 * dim + 1):
        missing = sym.get_missing_dists(grid, o)
        for k in range(n):
            fi = out['f'][k].copy()
            for i in missing:
                fi[i] = fi[grid.idx_opposite[i]]
            rs = _evalf(sym.ex_rho(grid, 'fi', False, minimize_roundoff=True), _fi_subs(grid, fi))
            # equilibrium velocity node
            subs = _macro_subs(grid, rs, bc_v[k])
            r = _evalf(sym.ex_rho(grid, 'fi', False, missing_dir=o, minimize_roundoff=True), subs)
            eqv_rho[o - 1, k] = r
            subs = _macro_subs(grid, r, bc_v[k])
            eqv[o - 1, k] = [_evalf(e, subs) for e in eq_ro]
            # equilibrium density node
            subs = {'g0m0': rs, 'rho': rs, 'par_rho': bc_rho[k]}
            vv = [_evalf(sym.ex_velocity(grid, 'fi', d, cfg_ro, missing_dir=o, par_rho='par_rho'), subs) for d in range(dim)]
            eqd_v[o - 1, k] = vv
            subs = _macro_subs(grid, bc_rho[k] - 1.0, vv)
            eqd[o - 1, k] = [_evalf(e, subs) for e in eq_ro]
        # the regularized node on an equilibrium state, both formulations
        r0, u0 = 1.03, [0.04, -0.02, 0.03][:dim]
        for col, (c, eqx, shift) in enumerate(((cfg, eq_std, 0.0), (cfg_ro, eq_ro, 1.0))):
            subs = _macro_subs(grid, r0 - shift, u0)
            g = np.array([_evalf(e, subs) for e in eqx])            # f = feq exactly (shifted under the option)
            subs.update(_fi_subs(grid, g))
            fl = []
            for a in range(dim):
                for b in range(a, dim):
                    fl.append(_evalf(sym.ex_flux(grid, 'fi', a, b, c), subs) - _evalf(sym.ex_eq_flux(grid, a, b), subs))
            subs.update({'flux[%d]' % j: x for j, x in enumerate(fl)})
            res = np.array([_evalf(eqx[i], subs) + _evalf(reg[i], subs) for i in range(Q)])
            reg_defect[o - 1, col] = float(np.max(np.abs(res - g)))
    out.update(eqvel_drho=eqv_rho, eqvel_post=eqv, eqdens_v=eqd_v, eqdens_post=eqd, regvel_defect_std_vs_roundoff=reg_defect)
    return out


def turbulence_goldens(grid, rng, n=24):
    """--regularized and --subgrid=les-smagorinsky (reference lb_single.py:27-42), composed from the reference's own
    objects in the order of bgk_relaxation_preamble (relaxation_common.mako:205-237): equilibrium at (rho, v0) ->
    update_relaxation_time (:166-203: strain from ex_flux(d0) - ex_eq_flux, tau0 += (sqrt(tau0^2 + 36 C^2 sqrt(Q)) - tau0) / 2)
    -> regularisation (:228-236: d0 = feq + reglb_flux_tensor . flux, the flux of the populations as read) -> BGK_relaxate
    (relaxation.mako:124-132, omega = 1 / tau0 under the subgrid model) -> Guo term with the prefactor at tau0
    (sym_force.py:140-160; relaxation_common.mako:110-135)."""
    dim, Q = grid.dim, grid.Q
    cfg = _Cfg()
    rho = rng.uniform(0.9, 1.1, n)
    f = np.array([[float(w) for w in grid.weights]] * n) * rho[:, None]
    f = f * (1.0 + rng.uniform(-0.08, 0.08, (n, Q)))
    accel = rng.uniform(-1e-4, 1e-4, (n, dim))
    visc = np.array([0.001, 0.02])
    csmag = np.array([0.1, 0.17])
    eq = sym_equilibrium.bgk_equilibrium(grid, cfg)
    reg = sym.reglb_flux_tensor(grid)
    guo = sym_force.guo_external_force(grid, grid_num=0)
    pref_e = sym_force.guo_external_force_pref(grid, cfg, grid_num=0)
    pairs = [(a, b) for a in range(dim) for b in range(a, dim)]
    out = {'f': f, 'accel': accel, 'visc': visc, 'smagorinsky_const': csmag}
    shape = (len(visc), len(csmag), 2, n, Q)            # [viscosity][constant][without / with body force][sample]
    for name in ('reg_post', 'les_post', 'reg_les_post'):
        out[name] = np.zeros(shape)
    out['les_tau'] = np.zeros(shape[:-1])
    out['out_v'] = np.zeros((2, n, dim))
    for k in range(n):
        fs = _fi_subs(grid, f[k])
        fs.update(_fi_subs(grid, f[k], ptr='d0'))
        r = _evalf(sym.ex_rho(grid, 'fi', False), fs)
        fs2 = dict(fs)
        fs2.update({'g0m0': r, 'rho': r})
        u = np.array([_evalf(sym.ex_velocity(grid, 'fi', d, cfg), fs2) for d in range(dim)])
        for forced in (0, 1):
            v0 = u + (0.5 * accel[k] if forced else 0.0)
            out['out_v'][forced, k] = v0
            ms = _macro_subs(grid, r, v0)
            feq = np.array([_evalf(e, ms) for e in eq.expression])
            sub = dict(fs)
            sub.update(ms)
            flux = np.array([_evalf(sym.ex_flux(grid, 'd0', a, b, cfg), sub) - _evalf(sym.ex_eq_flux(grid, a, b), ms)
                             for a, b in pairs])
            fl = {'flux[%d]' % j: flux[j] for j in range(len(pairs))}
            d_reg = feq + np.array([_evalf(e, fl) for e in reg])
            strain = sum((2.0 if a != b else 1.0) * flux[j] ** 2 for j, (a, b) in enumerate(pairs))
            for vi, nu in enumerate(visc):
                tau = sym.relaxation_time(nu)
                for ci, c in enumerate(csmag):
                    tau_les = tau + 0.5 * (np.sqrt(tau * tau + 36.0 * c * c * np.sqrt(strain)) - tau)
                    out['les_tau'][vi, ci, forced, k] = tau_les
                    for name, d0, t0 in (('reg_post', d_reg, tau), ('les_post', f[k], tau_les), ('reg_les_post', d_reg, tau_les)):
                        post = d0 + (1.0 / t0) * (feq - d0)
                        if forced:
                            gs = dict(ms)
                            gs.update({'g0ea' + ch: accel[k, j] for j, ch in enumerate('xyz'[:dim])})
                            gs['tau0'] = t0
                            gs['pref'] = _evalf(pref_e, gs)
                            post = post + np.array([_evalf(g, gs) for g in guo])
                        out[name][vi, ci, forced, k] = post
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'tables':      # the lattice tables only (the .npz files stay as they are)
        with open(os.path.join(OUT, 'lattices.json'), 'w') as fh:
            json.dump({g.__name__: lattice_tables(g) for g in (sym.D2Q9, sym.D3Q19)}, fh, indent=1, sort_keys=True)
        print('wrote lattices.json')
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'turbulence':     # --regularized / --subgrid=les-smagorinsky only
        rng_t = np.random.RandomState(61)
        for grid in (sym.D2Q9, sym.D3Q19):
            np.savez_compressed(os.path.join(OUT, 'arith_reg_les_%s.npz' % grid.__name__), **turbulence_goldens(grid, rng_t))
            print('wrote regularized / Smagorinsky goldens for', grid.__name__)
        return
    rng_sc = np.random.RandomState(777)
    for grid in (sym.D2Q9, sym.D3Q19):
        np.savez_compressed(os.path.join(OUT, 'shan_chen_%s.npz' % grid.__name__), **shan_chen_goldens(grid, rng_sc))
        print('wrote Shan-Chen goldens for', grid.__name__)
    if len(sys.argv) > 1 and sys.argv[1] == 'sc':
        return
    rng_ro = np.random.RandomState(4242)
    for grid in (sym.D2Q9, sym.D3Q19):
        np.savez_compressed(os.path.join(OUT, 'arith_ro_bc_%s.npz' % grid.__name__), **roundoff_bc_goldens(grid, rng_ro))
        print('wrote --minimize_roundoff boundary-node goldens for', grid.__name__)
    if len(sys.argv) > 1 and sys.argv[1] == 'ro_bc':
        return
    rng = np.random.RandomState(20260926)
    tables = {}
    for grid in (sym.D2Q9, sym.D3Q19):
        tables[grid.__name__] = lattice_tables(grid)
        ar = arithmetic_goldens(grid, rng)
        np.savez_compressed(os.path.join(OUT, 'arith_%s.npz' % grid.__name__), **ar)
        print('wrote arithmetic goldens for', grid.__name__)
    with open(os.path.join(OUT, 'lattices.json'), 'w') as fh:
        json.dump(tables, fh, indent=1, sort_keys=True)
    print('wrote lattices.json')


if __name__ == '__main__':
    main()
