"""The reference's face pack / unpack kernels under their own names and argument lists (SURVEY.md §2.3;
kernel_utils.mako:476-953, bound at subdomain_runner.py:1160-1290): Collect / DistributeContinuousData,
...ContinuousDataWithSwap (AA modules), Collect / DistributeContinuousMacroData -- (dist, face, base_gx, base_other,
max_lx, max_other, buffer) in 3-D, (dist, face, base_gx, max_lx, buffer) in 2-D.  The expected buffers are restated
here with numpy from the template text: layer = lat_linear / lat_linear_macro / lat_linear_dist / lat_linear_with_swap
(subdomain_runner.py:486-510), populations = get_interblock_dists(grid, normal(face)) ascending, buffer [k][other][x]."""
import numpy as np
import pytest

from sailfish_amd import sym
from sailfish_amd.box import BoxSim, make_box_desc

pytestmark = pytest.mark.gpu
Y_LOW, Y_HIGH, Z_LOW, Z_HIGH = 2, 3, 4, 5


def _backend():
    from sailfish_amd.backend_hip import HIPBackend

    class Opt(object):
        pass
    return HIPBackend(Opt(), 0)


def _dists(grid, face):
    axis, sign = face >> 1, (-1 if face % 2 == 0 else 1)
    return [i for i in range(1, grid.Q) if grid.basis[i][axis] == sign]


def _layers(lat, face):
    """(lat_linear, lat_linear_macro, lat_linear_dist, lat_linear_with_swap)[face], envelope 1"""
    low = face % 2 == 0
    return ((0 if low else lat - 1), (1 if low else lat - 2), (lat - 2 if low else 1), (lat - 1 if low else 0))


def _node_box(arr, face, layer, base_gx, base_other, nx, nother):
    """arr[..., z, y, x] -> [..., other, x] of the face layer"""
    if face >> 1 == 1:
        return arr[..., base_other:base_other + nother, layer, base_gx:base_gx + nx]
    return arr[..., layer, base_other:base_other + nother, base_gx:base_gx + nx]


@pytest.mark.parametrize('face', [Y_LOW, Y_HIGH, Z_LOW, Z_HIGH])
@pytest.mark.parametrize('precision', ['single', 'double'])
def test_reference_face_kernels_3d(face, precision):
    b = _backend()
    grid = sym.D3Q19
    size = (20, 9, 7)
    desc = make_box_desc(grid, size, precision=precision, access_pattern='AA', visc=0.02)
    sim = BoxSim(b, desc)
    rng = np.random.RandomState(face)
    f = rng.rand(19, *sim.shape).astype(sim.dtype)
    sim.set_dist(f)
    lat = [desc.lat_nx, desc.lat_ny, desc.lat_nz]
    axis = face >> 1
    base_gx, nx = 2, 15
    nother_full = lat[3 - axis]          # the other in-plane axis (z for y faces, y for z faces)
    base_other, nother = 1, nother_full - 3
    dists = _dists(grid, face)
    ll, ll_macro, ll_dist, ll_swap = _layers(lat[axis], face)
    isz = sim.dtype().itemsize
    nbuf = len(dists) * nother * nx
    dbuf = b.alloc_buf(size=nbuf * isz)
    host = np.zeros(nbuf, dtype=sim.dtype)
    args = [face, base_gx, base_other, nx, nother * len(dists)]

    def run(name, ptr):
        k = b.get_kernel(sim.module, name, (64,), [ptr] + args + [dbuf], 'PiiiiiP')
        b.run_kernel(k, None, sim.stream)
        sim.sync()

    # ---- collect: ghost layer, same slots / first real layer, opposite slots
    for name, layer, slots in (('CollectContinuousData', ll, dists),
                               ('CollectContinuousDataWithSwap', ll_macro, [grid.idx_opposite[d] for d in dists])):
        run(name, sim.gpu_dist[0])
        b.from_buf(dbuf, host)
        want = np.stack([_node_box(f[q], face, layer, base_gx, base_other, nx, nother) for q in slots])
        assert np.array_equal(host.reshape(want.shape), want), name
    # ---- distribute: the values land in the layer on the far side, nothing else changes
    for name, layer, slots in (('DistributeContinuousData', ll_dist, dists),
                               ('DistributeContinuousDataWithSwap', ll_swap, [grid.idx_opposite[d] for d in dists])):
        sim.set_dist(f)
        payload = rng.rand(len(dists), nother, nx).astype(sim.dtype)
        b.to_buf(dbuf, np.ascontiguousarray(payload).reshape(-1))
        run(name, sim.gpu_dist[0])
        got = sim.get_dist()
        want = f.copy()
        for k_, q in enumerate(slots):
            _node_box(want[q], face, layer, base_gx, base_other, nx, nother)[...] = payload[k_]
        assert np.array_equal(got, want), name
    # ---- macroscopic fields: first real layer -> buffer -> ghost layer, every value delivered (infinities too)
    fld = rng.rand(*sim.shape).astype(sim.dtype)
    fld[1, 1, 3] = np.inf
    sim.rho[...] = fld
    b.to_buf(sim.gpu_rho)
    margs = [face, base_gx, base_other, nx, nother]
    k = b.get_kernel(sim.module, 'CollectContinuousMacroData', (64,), [sim.gpu_rho] + margs + [dbuf], 'PiiiiiP')
    b.run_kernel(k, None, sim.stream)
    sim.sync()
    b.from_buf(dbuf, host)
    want = _node_box(fld, face, ll_macro, base_gx, base_other, nx, nother)
    assert np.array_equal(host[:want.size].reshape(want.shape), want)
    k = b.get_kernel(sim.module, 'DistributeContinuousMacroData', (64,), [sim.gpu_rho] + margs + [dbuf], 'PiiiiiP')
    b.run_kernel(k, None, sim.stream)
    sim.sync()
    b.from_buf(sim.gpu_rho)
    after = fld.copy()
    _node_box(after, face, ll, base_gx, base_other, nx, nother)[...] = want
    assert np.array_equal(sim.rho, after)
    b.free_buf(dbuf)
    sim.release()


@pytest.mark.parametrize('face', [Y_LOW, Y_HIGH])
def test_reference_face_kernels_2d(face):
    b = _backend()
    grid = sym.D2Q9
    desc = make_box_desc(grid, (24, 11), access_pattern='AA', visc=0.02)
    sim = BoxSim(b, desc)
    rng = np.random.RandomState(10 + face)
    f = rng.rand(9, *sim.shape).astype(sim.dtype)
    sim.set_dist(f)
    dists = _dists(grid, face)
    ll, ll_macro, ll_dist, ll_swap = _layers(desc.lat_ny, face)
    base_gx, nx = 3, 17
    dbuf = b.alloc_buf(size=len(dists) * nx * 4)
    host = np.zeros(len(dists) * nx, dtype=np.float32)
    for name, layer, slots in (('CollectContinuousData', ll, dists),
                               ('CollectContinuousDataWithSwap', ll_macro, [grid.idx_opposite[d] for d in dists])):
        k = b.get_kernel(sim.module, name, (64,), [sim.gpu_dist[0], face, base_gx, nx * len(dists), dbuf], 'PiiiP')
        b.run_kernel(k, None, sim.stream)
        sim.sync()
        b.from_buf(dbuf, host)
        want = np.stack([f[q, 0, layer, base_gx:base_gx + nx] for q in slots])
        assert np.array_equal(host.reshape(want.shape), want), name
    for name, layer, slots in (('DistributeContinuousData', ll_dist, dists),
                               ('DistributeContinuousDataWithSwap', ll_swap, [grid.idx_opposite[d] for d in dists])):
        sim.set_dist(f)
        payload = rng.rand(len(dists), nx).astype(np.float32)
        b.to_buf(dbuf, np.ascontiguousarray(payload).reshape(-1))
        k = b.get_kernel(sim.module, name, (64,), [sim.gpu_dist[0], face, base_gx, nx * len(dists), dbuf], 'PiiiP')
        b.run_kernel(k, None, sim.stream)
        sim.sync()
        want = f.copy()
        for k_, q in enumerate(slots):
            want[q, 0, layer, base_gx:base_gx + nx] = payload[k_]
        assert np.array_equal(sim.get_dist(), want), name
    # macro: (field, base_gx, max_lx, gy, buffer)
    fld = rng.rand(*sim.shape).astype(np.float32)
    sim.rho[...] = fld
    b.to_buf(sim.gpu_rho)
    k = b.get_kernel(sim.module, 'CollectContinuousMacroData', (64,), [sim.gpu_rho, base_gx, nx, 4, dbuf], 'PiiiP')
    b.run_kernel(k, None, sim.stream)
    k = b.get_kernel(sim.module, 'DistributeContinuousMacroData', (64,), [sim.gpu_rho, base_gx, nx, 7, dbuf], 'PiiiP')
    b.run_kernel(k, None, sim.stream)
    sim.sync()
    b.from_buf(sim.gpu_rho)
    want = fld.copy()
    want[0, 7, base_gx:base_gx + nx] = fld[0, 4, base_gx:base_gx + nx]
    assert np.array_equal(sim.rho, want)
    b.free_buf(dbuf)
    sim.release()


def test_swap_kernels_exist_for_the_in_place_pattern_only():
    b = _backend()
    desc = make_box_desc(sym.D3Q19, (8, 6, 5), access_pattern='AB')
    sim = BoxSim(b, desc)
    with pytest.raises(b.FatalError):
        b.get_kernel(sim.module, 'CollectContinuousDataWithSwap', (64,), [sim.gpu_dist[0], 2, 1, 1, 4, 5, sim.gpu_rho], 'PiiiiiP')
    sim.release()


def test_face_box_that_leaves_the_arrays_is_refused():
    """base_other + rows (and the 12-direction limit of the box form) are checked before anything is
    written; a bad max_other used to write out of bounds from DistributeContinuousData."""
    from sailfish_amd.backend_hip import HIPFatalError
    b = _backend()
    desc = make_box_desc(sym.D3Q19, (20, 9, 7), precision='single', access_pattern='AA', visc=0.02)
    sim = BoxSim(b, desc)
    dbuf = b.alloc_buf(size=1 << 20)
    rows_y, rows_z = desc.lat_ny, desc.lat_nz
    for face, rows in ((Z_LOW, rows_y), (Y_HIGH, rows_z)):
        ok = b.get_kernel(sim.module, 'DistributeContinuousData', (64,), [sim.gpu_dist[0], face, 1, 1, 10, (rows - 1) * 5, dbuf], 'PiiiiiP')
        b.run_kernel(ok, None, sim.stream)
        sim.sync()
        bad = b.get_kernel(sim.module, 'DistributeContinuousData', (64,), [sim.gpu_dist[0], face, 1, 1, 10, rows * 5, dbuf], 'PiiiiiP')
        with pytest.raises(HIPFatalError, match='outside the subdomain'):
            b.run_kernel(bad, None, sim.stream)
    many = b.get_kernel(sim.module, 'CollectContinuousData', (64,), [sim.gpu_dist[0], dbuf, (1 << 13) - 1, 0, 1, 4, desc.arr_nx, 2], 'PPiiiiii')
    with pytest.raises(HIPFatalError, match='at most 12 directions'):
        b.run_kernel(many, None, sim.stream)


@pytest.mark.parametrize('case,pair,high', [('3d_z2_open', '0->1', True), ('3d_z2_open', '1->0', False),
                                            ('3d_y3_periodic_y', '1->2', True), ('3d_y3_periodic_y', '2->0', True),
                                            ('3d_y3_periodic_y', '0->2', False), ('2d_y2_periodic_y', '0->1', True),
                                            ('2d_y2_periodic_y', '0->1', False), ('2d_y2_periodic_y', '1->0', True)])
def test_reference_face_kernels_move_what_the_reference_connections_transfer(case, pair, high):
    """The expected values above are restated from the template text in this file; this test pins the same kernels to a
    fixture that comes from the reference's own objects: tests/golden/connections.json = its LBConnection objects expanded
    into sets {(population, receiver node)} per ordered pair of subdomains (tools/capture_connections.py).  Collect on the
    sender + Distribute on the receiver, both under the reference's argument lists, over the interior of the face (where
    every population of the face has its source inside the sender): the slots that change on the receiver are exactly
    the fixture's pairs there, and each carries what the sender held for that population one layer beyond its face."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'connections.json')))[case]
    dim = len(gold['gsize'])
    grid = sym.D2Q9 if dim == 2 else sym.D3Q19
    s_id, r_id = [int(v) for v in pair.split('->')]
    (s_loc, s_size), (r_loc, r_size) = gold['boxes'][s_id], gold['boxes'][r_id]
    axis = [a for a in range(dim) if s_loc[a] != r_loc[a]][0]
    # high: the populations leave through the sender's high face (the receiver sits above it, or below across the periodic
    # seam); a periodic ring of two connects the same pair through both faces
    n_axis = gold['gsize'][axis]
    above = (s_loc[axis] + s_size[axis]) % n_axis == r_loc[axis] and (gold['periodic'][axis] or s_loc[axis] + s_size[axis] < n_axis)
    below = (r_loc[axis] + r_size[axis]) % n_axis == s_loc[axis] and (gold['periodic'][axis] or r_loc[axis] + r_size[axis] < n_axis)
    assert above if high else below
    face = 2 * axis + (1 if high else 0)
    b = _backend()
    sims = []
    for size in (s_size, r_size):
        desc = make_box_desc(grid, tuple(size), precision='double', access_pattern='AB', visc=0.02)
        sims.append(BoxSim(b, desc))
    snd, rcv = sims
    fs = np.arange(grid.Q * snd.nodes, dtype=np.float64).reshape((grid.Q,) + snd.shape) + 1.0     # every slot its own value
    fr = np.full((grid.Q,) + rcv.shape, -1.0)
    snd.set_dist(fs, which=0)
    rcv.set_dist(fr, which=0)
    dists = _dists(grid, face)
    # interior of the face: real nodes 1 .. n - 2 along the in-plane axes (array coordinates 2 .. n - 1)
    inplane = [a for a in range(dim) if a != axis]
    nx = s_size[0] - 2
    if dim == 3:
        other = inplane[1] if inplane[0] == 0 else inplane[0]
        nother = s_size[other] - 2
        args = [face, 2, 2, nx, nother * len(dists)]
        fmt = 'PiiiiiP'
    else:
        nother = 1
        args = [face, 2, nx * len(dists)]
        fmt = 'PiiiP'
    dbuf = b.alloc_buf(size=len(dists) * nother * nx * 8)
    for sim, name in ((snd, 'CollectContinuousData'), (rcv, 'DistributeContinuousData')):
        k = b.get_kernel(sim.module, name, (64,), [sim.gpu_dist[0]] + args + [dbuf], fmt)
        b.run_kernel(k, None, sim.stream)
        sim.sync()
    got = rcv.get_dist(which=0)
    changed = np.argwhere(got != fr)
    # (q, array z, array y, array x) -> the fixture's (q, real x, real y[, real z])
    moved = set()
    for idx in changed:
        q, coords = int(idx[0]), [int(c) - 1 for c in idx[1:]][::-1]       # x, y, z real coordinates
        moved.add(tuple([q] + coords[:dim]))
    lo = [1] * dim
    hi = [n - 2 for n in r_size]
    layer_r = 0 if high else r_size[axis] - 1              # the receiver's first / last real layer
    want = set(tuple(t) for t in gold['pairs'][pair]
               if t[1 + axis] == layer_r and grid.basis[t[0]][axis] == (1 if high else -1) and
               all(lo[a] <= t[1 + a] <= hi[a] for a in inplane))
    assert want and moved == want, (sorted(moved ^ want)[:8], len(moved), len(want))
    assert all(len([t for t in want if t[1:] == node]) == len(dists) for node in set(t[1:] for t in want))
    # the values: population q of the receiver's node = what the sender held for q in the layer beyond its face
    layer_s = snd.shape[3 - 1 - axis] - 1 if high else 0                   # the sender's ghost layer of that face
    for t in want:
        q, pos = t[0], list(t[1:]) + [0] * (3 - dim)
        r_idx = [pos[2] + 1 if dim == 3 else 0, pos[1] + 1, pos[0] + 1]
        s_idx = list(r_idx)
        s_idx[2 - axis] = layer_s
        assert got[(q,) + tuple(r_idx)] == fs[(q,) + tuple(s_idx)]
    b.free_buf(dbuf)
    for sim in sims:
        sim.release()
