"""NTDoNothing and NTSlip through the host stack (set_node -> orientation -> encoder -> type table -> module descriptor)
with the oracle doing the arithmetic: what the reference's own test of the node type asserts (tests/gpu/do_nothing_node.py:
an in-place run with NTDoNothing outlets gives the fields of the two-copy run), the 1-vs-N subdomain equivalence of its
regression tests, and the physics of a slip wall.  No GPU."""
import numpy as np
import pytest

from sailfish_amd import hipabi, node_type
from tests import _open_sims as S
from tests._oracle_group import OracleGroup

GEO = {2: 'EqualSubdomainsGeometry2D', 3: 'EqualSubdomainsGeometry3D'}


def _run(sim, dim, steps, **cfg):
    og = OracleGroup(sim, dim, GEO[dim], cfg)
    og.run(steps, save_last=True)
    return og


def test_kernel_kind_of_the_do_nothing_node_depends_on_the_access_pattern():
    assert node_type.hip_kind(node_type.NTDoNothing, 'AA') == hipabi.SLF_NK_DO_NOTHING
    assert node_type.hip_kind(node_type.NTDoNothing, 'AB') == hipabi.SLF_NK_FLUID      # node_type.py:296-307 of the reference
    assert node_type.hip_kind(node_type.NTSlip, 'AA') == node_type.hip_kind(node_type.NTSlip, 'AB') == hipabi.SLF_NK_SLIP
    assert node_type.hip_kind(node_type.NTGradFreeflow, 'AB') is None
    for pattern, kind in (('AA', hipabi.SLF_NK_DO_NOTHING), ('AB', hipabi.SLF_NK_FLUID)):
        og = OracleGroup(S.OpenChannelSim, 2, GEO[2], dict(lat_nx=16, lat_ny=10, visc=0.05, access_pattern=pattern))
        desc = og.subs[0].desc
        kinds = [desc.type_kind[i] for i in range(desc.n_types)]
        assert (hipabi.SLF_NK_DO_NOTHING in kinds) == (pattern == 'AA'), kinds
        assert kind in kinds


@pytest.mark.parametrize('dim', [2, 3])
def test_do_nothing_outlet_in_place_is_the_two_copy_run_for_any_decomposition(dim):
    if dim == 2:
        sim, base, steps, comp = S.OpenChannelSim, dict(lat_nx=40, lat_ny=24, visc=0.05), 61, 'v0'
        splits = ((1, 'x'), (2, 'y'), (2, 'x'), (3, 'y'))
    else:
        sim, base, steps, comp = S.OpenDuctSim, dict(lat_nx=10, lat_ny=12, lat_nz=20, visc=0.05, periodic_x=True), 40, 'v2'
        splits = ((1, 'x'), (2, 'z'), (2, 'y'), (2, 'x'))
    ref = None
    for pattern in ('AB', 'AA'):
        for nsub, axis in splits:
            og = _run(sim, dim, steps, access_pattern=pattern, subdomains=nsub, conn_axis=axis, **base)
            got = [og.merged('rho'), og.merged(comp)]
            if ref is None:
                ref = got
                wet = np.isfinite(ref[0]) & (ref[0] != 0)
                assert np.all(np.isfinite(ref[1][wet]))
                continue
            for a, b in zip(ref, got):
                assert np.array_equal(a[wet], b[wet]), (pattern, nsub, axis)
    # the flow leaves through the outlet
    assert np.all(ref[1][..., 2:-2, -1] > 0.02) if dim == 2 else np.all(ref[1][-1, 2:-2, :] > 0.02)


@pytest.mark.parametrize('pattern', ['AA', 'AB'])
def test_slip_walls_leave_a_uniform_flow_alone(pattern):
    """Specular reflection returns the tangential momentum: a uniform flow along two slip walls stays what it is (between
    bounce-back walls the same flow decays from the walls inwards)."""
    class Sub(S.SlipChannelSubdomain):
        u0 = 0.03

    class Sim(S.SlipChannelSim):
        subdomain = Sub
        accel = 0.0

    for nsub, axis in ((1, 'x'), (2, 'y')):
        og = _run(Sim, 2, 300, lat_nx=20, lat_ny=16, visc=0.05, periodic_x=True, access_pattern=pattern, precision='double',
                  subdomains=nsub, conn_axis=axis)
        vx, vy = og.merged('v0'), og.merged('v1')
        assert np.max(np.abs(vx[1:-1] - 0.03)) < 1e-14
        assert np.max(np.abs(vy[1:-1])) < 1e-14


def test_slip_walls_in_three_dimensions_equal_for_any_decomposition():
    ref = None
    for pattern in ('AB', 'AA'):
        for nsub, axis in ((1, 'x'), (2, 'y'), (2, 'x')):
            og = _run(S.SlipDuctSim, 3, 60, lat_nx=12, lat_ny=10, lat_nz=8, visc=0.05, periodic_x=True, periodic_z=True,
                      force_implementation='guo', access_pattern=pattern, subdomains=nsub, conn_axis=axis)
            vx = og.merged('v0')
            if ref is None:
                ref = vx
                assert 0.9 * 60e-5 < vx[:, 1:-1].min() and vx[:, 1:-1].max() < 60e-5
            assert np.array_equal(vx, ref), (pattern, nsub, axis)
