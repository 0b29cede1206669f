"""Host logic of the slab driver (sailfish_amd/slab.py: SlabPlan) on the CPU: two oracle slabs that exchange exactly
the node boxes the plan describes -- expanded to index lists -- must reproduce the single box bit for bit, along x, y
and z, for the push (AB, odd AA) and the opposite-slot (even AA) halo flavours."""
import numpy as np
import pytest

from sailfish_amd import sym
from sailfish_amd.box import make_box_desc
from sailfish_amd.slab import AXES, SlabPlan
from tests._oracle_box import OracleBox, synthetic_fields


def _flat(ob, dist):
    """The [Q, stride] buffer behind an oracle distribution view."""
    return dist.base.reshape(-1) if dist.base is not None else dist.reshape(-1)


@pytest.mark.parametrize('axis', ['x', 'y', 'z'])
@pytest.mark.parametrize('pattern', ['AA', 'AB'])
def test_two_oracle_slabs_equal_one_box(axis, pattern):
    grid, a = sym.D3Q19, AXES[axis]
    n = [12, 7, 6]
    whole = list(n)
    whole[a] *= 2
    steps = 7
    rho, v = synthetic_fields(tuple(whole), 3)
    fused = [1, 1, 1]
    ref_desc = make_box_desc(grid, tuple(whole), access_pattern=pattern, visc=0.02, periodic_fused=fused)
    ref = OracleBox(ref_desc, periodic=(True, True, True))
    ref.set_fields(rho, v)
    ref.initial_conditions()
    ref.run(steps, save_last=False)

    part_fused = list(fused)
    part_fused[a] = 0
    slabs, plans = [], []
    for r in range(2):
        desc = make_box_desc(grid, tuple(n), access_pattern=pattern, visc=0.02, periodic_fused=part_fused)
        per = [True, True, True]
        per[a] = False
        ob = OracleBox(desc, periodic=tuple(per))
        sl = [slice(None)] * 3
        sl[2 - a] = slice(r * n[a], (r + 1) * n[a])
        ob.set_fields(rho[tuple(sl)], [c[tuple(sl)] for c in v])
        ob.initial_conditions()
        slabs.append(ob)
        plans.append(SlabPlan(grid, desc, a))
    assert plans[0].count == 5 * plans[0].ncols * plans[0].nrows
    for it in range(steps):
        swap = pattern == 'AA' and (it & 1) == 0
        for ob in slabs:
            ob.step()
        out = [ob.dist[0] if ob.aa else ob.dist[ob.iteration & 1] for ob in slabs]
        lists = [[p.index_list(b) for b in p.boxes(swap)] for p in plans]      # send_up, send_down, recv_low, recv_high
        bufs = [[_flat(slabs[r], out[r])[lists[r][j]].copy() for j in (0, 1)] for r in range(2)]
        for r in range(2):
            o = 1 - r                       # ring of two: the other slab is both neighbours
            _flat(slabs[o], out[o])[lists[o][2]] = bufs[r][0]       # my send_up   -> its recv_low
            _flat(slabs[o], out[o])[lists[o][3]] = bufs[r][1]       # my send_down -> its recv_high
    got = np.concatenate([ob.real_view(ob.current_dist()) for ob in slabs], axis=3 - a)
    assert np.array_equal(got, ref.real_view(ref.current_dist()))


def test_regions_cover_every_row_once():
    grid = sym.D3Q19
    for axis in 'xyz':
        desc = make_box_desc(grid, (16, 9, 8), periodic_fused=[0, 0, 0])
        p = SlabPlan(grid, desc, AXES[axis])
        bnd, bulk = p.regions()
        seen = np.zeros((8 + 2, 9 + 2), dtype=int)
        for y0, y1, z0, z1 in bnd + [bulk]:
            seen[z0:z1, y0:y1] += 1
        assert np.all(seen[1:-1, 1:-1] == 1) and seen.sum() == 9 * 8
        assert (len(bnd) == 0) == (axis == 'x')
