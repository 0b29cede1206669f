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
