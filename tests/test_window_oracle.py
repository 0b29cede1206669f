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


def _cut(global_box, desc_l, axis, r, n_l, copies, ghost=np.nan):
    """The arrays slab r of the global oracle box would hold: real nodes copied, ghost layers filled with `ghost`
    (a seam window must not look at them)."""
    out = []
    for c in copies:
        g = global_box.real_view(global_box.dist[c])
        sl = [slice(None)] * 4
        sl[3 - axis] = slice(r * n_l, (r + 1) * n_l)
        loc = np.full((19, desc_l.arr_nz, desc_l.arr_ny, desc_l.arr_nx), ghost, dtype=np.float32)
        loc[:, 1:desc_l.lat_nz - 1, 1:desc_l.lat_ny - 1, 1:desc_l.lat_nx - 1] = g[tuple(sl)]
        out.append(np.ascontiguousarray(loc))
    return out


@pytest.mark.parametrize('axis', [0, 1, 2])
@pytest.mark.parametrize('pattern', ['AA', 'AB'])
@pytest.mark.parametrize('start', [4, 5])
def test_seam_windows_reach_into_the_neighbouring_slabs(axis, pattern, start):
    """window.SeamCheck (bench.py's validation of N > 1 runs): a periodic box cut into two slabs; the windows of one
    slab take three layers of the other through swap() and reproduce the undivided oracle run on the seam layers."""
    from oracle.window import SeamCheck
    size_l = [12, 10, 14]
    size_g = list(size_l)
    size_g[axis] *= 2
    kw = dict(model='bgk', precision='single', access_pattern=pattern, visc=0.02)
    desc_g = make_box_desc(sym.D3Q19, tuple(size_g), periodic_fused=[1, 1, 1], **kw)
    fused_l = [1, 1, 1]
    fused_l[axis] = 0
    desc_l = make_box_desc(sym.D3Q19, tuple(size_l), periodic_fused=fused_l, **kw)
    o = OracleBox(desc_g, periodic=(True, True, True))
    rho, v = synthetic_fields(tuple(size_g), 3)
    o.set_fields(rho, v)
    o.initial_conditions()
    o.run(start, save_last=False)
    copies = range(len(o.dist))
    before = [_cut(o, desc_l, axis, r, size_l[axis], copies) for r in (0, 1)]
    E = 3
    n = size_l[axis]

    def edges(arrs, lo):
        sl = [slice(None)] * 4
        sl[3 - axis] = slice(1, 1 + E) if lo else slice(n - E + 1, n + 1)
        return [np.ascontiguousarray(a[tuple(sl)]) for a in arrs]

    zs = [1, 7, size_l[2]]
    checks = []
    for r in (0, 1):
        other = before[1 - r]

        def swap(low, high, other=other):
            # a ring of two: both neighbours are the other slab; z: whole layers, x / y: per window [19, W, ...] blocks
            if axis == 2:
                return edges(other, False), edges(other, True)
            down, up = [], []
            for w in chk.windows:
                pl = w.planes
                down.append([np.ascontiguousarray(e[:, pl]) for e in edges(other, False)])
                up.append([np.ascontiguousarray(e[:, pl]) for e in edges(other, True)])
            return down, up
        chk = SeamCheck(HostMemory, desc_l, zs, [a.ctypes.data for a in before[r]], desc_l.arr_nx * desc_l.arr_ny * desc_l.arr_nz,
                        None, axis, swap)
        chk.seed(o.iteration)
        checks.append(chk)
    o.run(2, save_last=False)
    after = [_cut(o, desc_l, axis, r, n, copies) for r in (0, 1)]
    for r, chk in enumerate(checks):
        chk.advance(2, save_last=False)
        chk.dist_addrs = [a.ctypes.data for a in after[r]]
        res = chk.compare(fields=False)
        assert res['dist_exact'], (r, res)
        assert res['compared_values'] == 19 * len(zs) * size_l[0] * size_l[1]
    # and a value spoilt on the seam is seen
    cur = 0 if pattern == 'AA' else (o.iteration & 1)
    idx = [5, zs[0], 3, 3]
    idx[3 - axis] = 1                                        # first real layer along the split axis
    after[0][cur][tuple(idx)] += np.float32(1e-6)
    assert not checks[0].compare(fields=False)['dist_exact']


@pytest.mark.parametrize('axis', [0, 1, 2])
@pytest.mark.parametrize('pattern', ['AA', 'AB'])
@pytest.mark.parametrize('start', [4, 5])
def test_global_windows_merge_the_slabs_of_every_rank(axis, pattern, start):
    """window.GlobalCheck (bench.py's second validation of N > 1 runs; BASELINE config 4: eight subdomains): a periodic
    box cut into three slabs; the shares the ranks hand in are merged into windows of the UNDIVIDED box, which reproduce
    the undivided oracle run on whole planes across every seam and the wrap; a value spoilt anywhere on a sampled plane
    -- in any slab -- is seen."""
    from oracle.window import GlobalCheck
    world = 3
    size_l = [12, 10, 8]
    size_g = list(size_l)
    size_g[axis] *= world
    kw = dict(model='bgk', precision='single', access_pattern=pattern, visc=0.02)
    desc_g = make_box_desc(sym.D3Q19, tuple(size_g), periodic_fused=[1, 1, 1], **kw)
    fused_l = [1, 1, 1]
    fused_l[axis] = 0
    desc_l = make_box_desc(sym.D3Q19, tuple(size_l), periodic_fused=fused_l, **kw)
    o = OracleBox(desc_g, periodic=(True, True, True))
    rho, v = synthetic_fields(tuple(size_g), 3)
    o.set_fields(rho, v)
    o.initial_conditions()
    o.run(start, save_last=False)
    copies = range(len(o.dist))
    n = size_l[axis]
    before = [_cut(o, desc_l, axis, r, n, copies) for r in range(world)]
    calls = {}

    def gather_for(rank):          # ranks 1 .. world-1 hand in first, rank 0 last: it gets the list
        count = [0]

        def gather(obj):
            slot = calls.setdefault(count[0], {})
            slot[rank] = obj
            count[0] += 1
            return [slot[r] for r in range(world)] if rank == 0 else None
        return gather

    zs = [1, size_g[2] // 2, size_g[2]]
    stride = desc_l.arr_nx * desc_l.arr_ny * desc_l.arr_nz
    checks = [GlobalCheck(HostMemory, desc_l, zs, [a.ctypes.data for a in before[r]], stride, axis, r, world, gather_for(r))
              for r in range(world)]
    for r in (1, 2, 0):
        checks[r].seed(o.iteration)
    o.run(2, save_last=False)
    after = [_cut(o, desc_l, axis, r, n, copies) for r in range(world)]
    cur = 0 if pattern == 'AA' else (o.iteration & 1)
    for spoil in (False, True):
        if spoil:
            idx = [7, 1, 2, 2]          # population 7, plane / row / column 1 or 2 of the LAST slab
            if axis == 2:
                idx[1] = size_l[2]      # global plane size_g[2] is its last local plane
            after[world - 1][cur][tuple(idx)] += np.float32(1e-6)
        res = None
        for r in (1, 2, 0):
            chk = checks[r]
            if not spoil:
                chk.advance(2)
            chk.pc.dist_addrs = [a.ctypes.data for a in after[r]]
            out = chk.compare()
            assert (out is None) == (r != 0)
            res = out if r == 0 else res
        assert res['dist_exact'] == (not spoil), res
        assert res['compared_values'] == 19 * len(zs) * size_g[0] * size_g[1]
