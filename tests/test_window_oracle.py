"""The plane-window checker (oracle/window.py) against a full oracle run: a window seeded from the full state and
advanced two steps reproduces the sampled plane of the full run bit for bit -- the property the full-size GPU parity
tests (tests/test_gpu_fullsize.py) and bench.py's validation leg rest on."""
import ctypes

import numpy as np
import pytest

from oracle.window import PlaneCheck
from sailfish_amd import sym
from sailfish_amd.box import make_box_desc
from tests import _geometry as geo
from tests._oracle_box import OracleBox, synthetic_fields


class HostMemory(object):
    """Stands in for the backend: 'device addresses' are host pointers."""

    @staticmethod
    def from_buf(addr, out):
        ctypes.memmove(out.ctypes.data, addr, out.nbytes)


def _box(case, pattern, model):
    size = (20, 9, 12)
    kw = dict(model=model, precision='single', access_pattern=pattern, visc=0.02)
    if case == 'fused':
        desc = make_box_desc(sym.D3Q19, size, periodic_fused=[1, 1, 1], **kw)
        return desc, None, (True, True, True), 'synthetic'
    if case == 'ghostpbc':
        desc = make_box_desc(sym.D3Q19, size, periodic_fused=[0, 0, 1], **kw)
        return desc, None, (True, True, True), 'synthetic'
    desc = make_box_desc(sym.D3Q19, size, fluid_only=False, type_kind=geo.TYPE_KIND, nt_bits=geo.NT_BITS,
                         node_params=[0.05, 0.0, 0.0], **kw)
    return desc, geo.cavity_3d(desc), (False, False, False), 'rest'


@pytest.mark.parametrize('case', ['fused', 'ghostpbc', 'cavity'])
@pytest.mark.parametrize('pattern', ['AA', 'AB'])
@pytest.mark.parametrize('model', ['bgk', 'mrt'])
@pytest.mark.parametrize('start', [4, 5])
def test_window_reproduces_full_run(case, pattern, model, start):
    desc, nmap, periodic, init = _box(case, pattern, model)
    o = OracleBox(desc, periodic=periodic, node_map=nmap)
    rho, v = synthetic_fields((20, 9, 12), 3)
    if init == 'rest':
        rho, v = np.ones_like(rho), [np.zeros_like(c) for c in v]
    o.set_fields(rho, v)
    o.initial_conditions()
    o.run(start, save_last=False)
    zs = [1, 2, 6, 11, 12] if case != 'ghostpbc' else [1, 6, 12]
    chk = PlaneCheck(HostMemory, desc, nmap, zs, [d.ctypes.data for d in o.dist], o.o.stride,
                     [o.rho.ctypes.data] + [c.ctypes.data for c in o.v])
    chk.seed(o.iteration)
    o.run(2, save_last=True)
    chk.advance(2, save_last=True)
    res = chk.compare()
    assert res['dist_exact'] and res['rho_err'] == 0.0 and res['v_abs_err'] == 0.0, res
    assert res['compared_values'] > 0.5 * 19 * res['nodes']


def test_window_detects_a_wrong_value():
    desc, nmap, periodic, _ = _box('fused', 'AA', 'bgk')
    o = OracleBox(desc, periodic=periodic)
    rho, v = synthetic_fields((20, 9, 12), 3)
    o.set_fields(rho, v)
    o.initial_conditions()
    o.run(2, save_last=False)
    chk = PlaneCheck(HostMemory, desc, None, [5], [o.dist[0].ctypes.data], o.o.stride)
    chk.seed(o.iteration)
    o.run(2, save_last=False)
    o.dist[0][7, 5, 4, 3] += np.float32(1e-6)
    chk.advance(2, save_last=False)
    res = chk.compare(fields=False)
    assert not res['dist_exact'] and res['dist_err'] > 0
