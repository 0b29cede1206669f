"""--node_addressing=indirect (reference subdomain_runner.py:829-878, kernel_common.mako:140-167) on the
CPU side: the active-node map of a wall-map geometry, the address table, and the oracle twin run with
sparse distribution arrays -- identical, bit for bit, to the dense run on every wet node."""
import numpy as np
import pytest

from sailfish_amd import hipabi
from tests import _host
from tests._oracle_group import OracleGroup

GEO = 'EqualSubdomainsGeometry3D'
BASE = dict(lat_nx=24, lat_ny=13, lat_nz=13, visc=0.05, periodic_x=True, grid='D3Q19')


def _group(addressing, **kw):
    cfg = dict(BASE, node_addressing=addressing, **kw)
    sim_cls = _host.load_sim_class('external_geometry', 'ExternalSimulation')
    return OracleGroup(sim_cls, 3, GEO, cfg)


def test_active_node_map_and_address_table():
    og = _group('indirect')
    sub = og.subs[0]
    r = sub.runner
    mask = r._subdomain.active_node_mask
    wall = r._subdomain._walls(*r._subdomain._get_mgrid_base(r.config))   # incl. ghosts, x wrapped (periodic)
    assert mask.shape == wall.shape
    fluid = ~wall
    assert np.all(mask[fluid])                       # every fluid node is active
    # a solid node is active iff it touches a fluid node (D3Q19 neighbourhood, wrapping like the reference)
    near = np.zeros_like(fluid)
    for e in r._sim.grid.basis[1:]:
        near |= np.roll(fluid, shift=(-e[2], -e[1], -e[0]), axis=(0, 1, 2))
    assert np.array_equal(mask, fluid | near)
    assert 0.3 < mask.mean() < 0.9                   # the point of the exercise: fewer slots than nodes
    addr = sub.addr.reshape(sub.o.shape)
    lat = addr[:mask.shape[0], :mask.shape[1], :mask.shape[2]]
    assert np.array_equal(lat != hipabi.SLF_INVALID_NODE, mask)
    slots = lat[mask]
    assert np.array_equal(slots, np.arange(mask.sum()))        # memory order
    assert np.all(addr[:, :, mask.shape[2]:] == hipabi.SLF_INVALID_NODE)   # x padding
    assert sub.desc.node_addressing == hipabi.SLF_ADDR_INDIRECT and sub.desc.fluid_only == 0
    assert sub.desc.dist_stride >= mask.sum() + 1 and sub.desc.dist_stride % 32 == 0


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('model', ['bgk', 'mrt'])
@pytest.mark.parametrize('nsub,axis', [(1, 'x'), (2, 'x'), (2, 'y')])
def test_indirect_equals_direct(pattern, model, nsub, axis):
    res = {}
    for addressing in ('direct', 'indirect'):
        og = _group(addressing, access_pattern=pattern, model=model, subdomains=nsub, conn_axis=axis)
        og.run(14, save_last=True)
        res[addressing] = (og.merged('rho'), og.merged('v0'), og.merged('dist'), og)
    rho_d, rho_i = res['direct'][0], res['indirect'][0]
    # the fluid nodes of the global domain (walls keep rho = 1 from the initial conditions, unused nodes too)
    og = res['direct'][3]
    wet = np.zeros(rho_d.shape, dtype=bool)
    for sub in og.subs:
        sp = sub.runner._spec
        sl = tuple(slice(o, o + n) for o, n in zip(reversed(sp.location), reversed(sp.size)))
        wet[sl] = sub.runner._subdomain.fluid_map()
    assert wet.sum() > 500
    assert np.array_equal(rho_d[wet], rho_i[wet])
    assert np.array_equal(res['direct'][1][wet], res['indirect'][1][wet])
    fd, fi = res['direct'][2], res['indirect'][2]
    assert np.array_equal(fd[:, wet], fi[:, wet])
    assert np.abs(res['indirect'][1][wet]).max() > 1e-6      # the body force has set the fluid in motion


# ---- binary Shan-Chen model (reference lb_binary.py:457-465: the nodes table leads every kernel's argument list) ----
def _sc_group(dim, size, addressing, nsub, axis, pattern, steps):
    from tests import _sc
    from tests._oracle_group import OracleNNGroup
    sim_cls, _ = _sc.make_wall_sim(dim)
    cfg = _sc.config(dim, size, pattern=pattern)
    cfg.update(periodic_y=False, subdomains=nsub, conn_axis=axis, node_addressing=addressing)
    g = OracleNNGroup(sim_cls, dim, 'EqualSubdomainsGeometry%dD' % dim, cfg)
    g.run(steps)
    return g


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('dim,size,nsub,axis', [(2, (24, 22), 1, 'x'), (3, (16, 20, 6), 1, 'x'), (3, (16, 20, 6), 2, 'x'),
                                                (3, (16, 20, 8), 2, 'z')])
def test_binary_shan_chen_indirect_equals_direct(pattern, dim, size, nsub, axis):
    """Sparse distribution arrays of both lattices, dense rho / phi / u: every field and every population of a wet
    node equals the dense run, bit for bit, on one and on two subdomains (population halo through translated links)."""
    d = _sc_group(dim, size, 'direct', nsub, axis, pattern, 9)
    i = _sc_group(dim, size, 'indirect', nsub, axis, pattern, 9)
    wet = np.zeros(tuple(reversed(d.subs[0].runner._global_size)), dtype=bool)
    for sub in d.subs:
        sp = sub.runner._spec
        sl = tuple(slice(o, o + n) for o, n in zip(reversed(sp.location), reversed(sp.size)))
        wet[sl] = sub.runner._subdomain.fluid_map()
    assert 0.3 < wet.mean() < 0.8
    for sub in i.subs:
        assert sub.indirect and sub.d1[0].shape == (sub.o.Q, sub.desc.dist_stride)
        assert sub.desc.dist_stride < 0.85 * np.prod(sub.runner._physical_size)
        assert 0.5 < sub.runner._subdomain.active_node_mask.mean() < 0.9
    for get in (lambda s: s.rho, lambda s: s.phi, lambda s: s.v[0], lambda s: s.v[1]):
        a, b = d.merged(get), i.merged(get)
        assert np.array_equal(a[wet], b[wet])
    for k in (0, 1):
        a = d.merged(lambda s: s.dense(s.current()[k]))
        b = i.merged(lambda s: s.dense(s.current()[k]))
        assert np.array_equal(a[:, wet], b[:, wet])
    rho = d.merged(lambda s: s.rho)[wet]
    assert np.isfinite(rho).all() and rho.std() > 1e-5
