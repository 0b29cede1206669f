"""1-vs-N subdomain equivalence on the CPU (reference regtest/subdomains/{2d_ldc,3d_ldc}.py run the
same simulation on 1 and on several subdomains and compare to 6 decimals, util.py:28-31; here the
comparison is bit-exact).  Geometry, descriptors and halo index lists come from the product's host
layer; the arithmetic is done by the oracle."""
import numpy as np
import pytest

from tests import _host
from tests._oracle_group import OracleGroup


def _run(module, sim, dim, geo, cfg, steps=12):
    sim_cls = _host.load_sim_class(module, sim)
    g = OracleGroup(sim_cls, dim, geo, cfg)
    g.run(steps, save_last=True)
    return g


def _compare(a, b):
    fa, fb = a.merged('dist'), b.merged('dist')
    assert np.array_equal(fa, fb, equal_nan=True), 'populations differ at %d places' % np.count_nonzero(
        ~((fa == fb) | (np.isnan(fa) & np.isnan(fb))))
    ra, rb = a.merged('rho'), b.merged('rho')
    assert np.array_equal(ra, rb, equal_nan=True)


LDC2 = dict(lat_nx=42, lat_ny=24, visc=0.02)
LDC3 = dict(lat_nx=20, lat_ny=14, lat_nz=12, visc=0.03)


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('nsub,axis', [(2, 'x'), (3, 'x'), (2, 'y'), (4, 'y')])
def test_ldc_2d_subdomains(pattern, nsub, axis):
    one = _run('ldc_2d', 'LDCSim', 2, 'EqualSubdomainsGeometry2D', dict(LDC2, access_pattern=pattern))
    many = _run('ldc_2d', 'LDCSim', 2, 'EqualSubdomainsGeometry2D',
                dict(LDC2, access_pattern=pattern, subdomains=nsub, conn_axis=axis))
    _compare(one, many)


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('nsub,axis', [(2, 'x'), (2, 'y'), (3, 'z')])
def test_ldc_3d_subdomains(pattern, nsub, axis):
    one = _run('ldc_3d', 'LDCSim', 3, 'EqualSubdomainsGeometry3D', dict(LDC3, access_pattern=pattern), steps=8)
    many = _run('ldc_3d', 'LDCSim', 3, 'EqualSubdomainsGeometry3D',
                dict(LDC3, access_pattern=pattern, subdomains=nsub, conn_axis=axis), steps=8)
    _compare(one, many)


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('nsub,axis', [(2, 'y'), (3, 'y'), (2, 'x')])
def test_periodic_channel_subdomains(pattern, fused, nsub, axis):
    """Force-driven channel: periodic along y.  Cutting along y makes the periodic images *other*
    subdomains (2 subdomains: both faces lead to the same neighbour); cutting along x combines a
    locally periodic axis (ghost-layer PBC or in-sweep wrap) with a decomposed one, edges included."""
    cfg = dict(lat_nx=20, lat_ny=24, visc=0.1, horizontal=False, stationary=False, drive='force', wall='halfbb',
               force_implementation='guo', access_pattern=pattern, hip_fused_periodic=fused)
    one = _run('poiseuille', 'PoiseuilleSim', 2, 'EqualSubdomainsGeometry2D', cfg)
    many = _run('poiseuille', 'PoiseuilleSim', 2, 'EqualSubdomainsGeometry2D',
                dict(cfg, subdomains=nsub, conn_axis=axis))
    _compare(one, many)


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('nsub,axis', [(2, 'z'), (3, 'z'), (2, 'x')])
def test_pipe_3d_subdomains(pattern, nsub, axis):
    cfg = dict(lat_nx=14, lat_ny=14, lat_nz=12, visc=0.1, flow_direction='z', stationary=False, drive='force',
               force_implementation='guo', access_pattern=pattern)
    one = _run('poiseuille_3d', 'PoiseuilleSim', 3, 'EqualSubdomainsGeometry3D', cfg, steps=8)
    many = _run('poiseuille_3d', 'PoiseuilleSim', 3, 'EqualSubdomainsGeometry3D',
                dict(cfg, subdomains=nsub, conn_axis=axis), steps=8)
    _compare(one, many)


def _block_geometry(dim, cuts):
    """Geometry class cutting the domain into a block grid: cuts = per-axis lists of cut positions (x first)."""
    from sailfish_amd.geo import LBGeometry2D, LBGeometry3D
    from sailfish_amd.subdomain import SubdomainSpec2D, SubdomainSpec3D
    base = LBGeometry2D if dim == 2 else LBGeometry3D

    class Blocks(base):
        def subdomains(self):
            g = [self.gx, self.gy] + ([self.gz] if dim == 3 else [])
            edges = [[0] + list(c) + [g[a]] for a, c in enumerate(cuts)]
            specs = []
            import itertools
            for idx in itertools.product(*[range(len(e) - 1) for e in edges]):
                loc = tuple(edges[a][i] for a, i in enumerate(idx))
                size = tuple(edges[a][i + 1] - edges[a][i] for a, i in enumerate(idx))
                specs.append((SubdomainSpec2D if dim == 2 else SubdomainSpec3D)(loc, size))
            return specs
    return Blocks


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('dim,cuts,size', [
    (2, [[9], [7]], (20, 16)),                    # 2 x 2 blocks of unequal size
    (2, [[6, 13], [8]], (20, 16)),                # 3 x 2
    (3, [[5], [4], [3]], (11, 9, 7)),             # 2 x 2 x 2: faces, edges and corners all cross subdomains
])
def test_block_decompositions_of_a_periodic_box(pattern, dim, cuts, size):
    """Every face, edge and corner neighbour incl. the periodic images is a different subdomain (or the
    subdomain itself across the seam): owner routing of the halo lists, 1 == N bit for bit, MRT."""
    from tests import _sc  # noqa: F401  (only for the import side effects of the helpers)
    from sailfish_amd.lb_single import LBFluidSim
    from sailfish_amd.subdomain import Subdomain2D, Subdomain3D

    class Box(Subdomain2D if dim == 2 else Subdomain3D):
        def boundary_conditions(self, *h):
            pass

        def initial_conditions(self, sim, *h):
            sim.rho[:] = 1.0 + 0.01 * np.sin(2 * np.pi * h[0] / self.gx) * np.cos(2 * np.pi * h[1] / self.gy)
            sim.vx[:] = 0.03 * np.sin(2 * np.pi * h[1] / self.gy)
            sim.vy[:] = 0.02 * np.cos(2 * np.pi * h[0] / self.gx)
            if dim == 3:
                sim.vz[:] = 0.01 * np.sin(2 * np.pi * (h[0] / self.gx + h[2] / self.gz))

    class Sim(LBFluidSim):
        subdomain = Box

    cfg = dict(lat_nx=size[0], lat_ny=size[1], periodic_x=True, periodic_y=True, visc=0.02, model='mrt',
               access_pattern=pattern, grid='D2Q9' if dim == 2 else 'D3Q19')
    if dim == 3:
        cfg.update(lat_nz=size[2], periodic_z=True)
    one = OracleGroup(Sim, dim, 'LBGeometry%dD' % dim, dict(cfg))
    many = OracleGroup(Sim, dim, _block_geometry(dim, cuts), dict(cfg))
    assert len(many.subs) == np.prod([len(c) + 1 for c in cuts])
    one.run(9, save_last=True)
    many.run(9, save_last=True)
    _compare(one, many)


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('addressing', ['direct', 'indirect'])
@pytest.mark.parametrize('nsub,vertical', [(2, False), (3, False), (2, True), (3, True)])
def test_cylinder_subdomains(pattern, addressing, nsub, vertical):
    """regtest/subdomains/2d_cylinder.py: an obstacle inside the flow, the channel periodic along the cut axis (the
    subdomains' periodic images are each other), lying and standing, dense and active-node storage."""
    size = dict(lat_nx=30, lat_ny=48) if vertical else dict(lat_nx=48, lat_ny=30)
    cfg = dict(size, visc=0.1, vertical=vertical, access_pattern=pattern, node_addressing=addressing,
               force_implementation='guo')
    one = _run('cylinder', 'CylinderSimulation', 2, 'EqualSubdomainsGeometry2D', cfg, steps=20)
    many = _run('cylinder', 'CylinderSimulation', 2, 'EqualSubdomainsGeometry2D',
                dict(cfg, subdomains=nsub, conn_axis='y' if vertical else 'x'), steps=20)
    _compare(one, many)
    v = one.merged('v1' if vertical else 'v0')
    assert np.nanmax(np.abs(v)) > 1e-5


@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('nsub,axis', [(2, 'x'), (2, 'y')])
def test_sphere_subdomains(pattern, nsub, axis):
    """regtest/subdomains/3d_sphere.py."""
    cfg = dict(lat_nx=30, lat_ny=15, lat_nz=18, visc=0.01, access_pattern=pattern, force_implementation='guo')
    one = _run('sphere_3d', 'SphereSimulation', 3, 'EqualSubdomainsGeometry3D', cfg, steps=10)
    many = _run('sphere_3d', 'SphereSimulation', 3, 'EqualSubdomainsGeometry3D',
                dict(cfg, subdomains=nsub, conn_axis=axis), steps=10)
    _compare(one, many)
