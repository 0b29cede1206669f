"""The oracle's full-slip and do-nothing nodes against what the reference pins for them.

* NTSlip (boundary.mako:837-855): the populations that swap are the reference's own `sym.slip_bb_swap_pairs` per
  orientation (tests/golden/lattices.json, dumped by tools/capture_goldens.py); a dry node: nothing else happens to it.
* NTDoNothing under the in-place pattern (boundary.mako:862-876): "the value from the previous time step is retained for
  all undefined distributions"; the reference's own check (tests/gpu/do_nothing_node.py) is that an in-place run with
  NTDoNothing outlets gives the fields of a two-copy run -- where those nodes are plain fluid nodes (node_type.py:296-307)
  and both copies start from the same equilibrium (lb_single.py:91-94).  Here the two runs are bit-identical.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle
from sailfish_amd import hipabi, sym
from sailfish_amd.box import make_box_desc
from tests import _geometry as geo
from tests._oracle_box import OracleBox

GRIDS = {'D2Q9': sym.D2Q9, 'D3Q19': sym.D3Q19}


@pytest.fixture(scope='module')
def tables(golden_dir):
    with open(os.path.join(golden_dir, 'lattices.json')) as fh:
        return json.load(fh)


@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
@pytest.mark.parametrize('precision', [4, 8])
def test_slip_node_swaps_the_reference_pairs(tables, name, precision):
    grid = GRIDS[name]
    desc = hipabi.make_desc(lattice=grid.slf_id, model=hipabi.SLF_BGK, precision=precision, access_pattern=hipabi.SLF_AB,
                            lat_nx=4, lat_ny=4, lat_nz=4 if grid.dim == 3 else 1, arr_nx=4, arr_ny=4,
                            arr_nz=4 if grid.dim == 3 else 1, tau=sym.relaxation_time(0.1), visc=0.1,
                            mrt_rates=sym.mrt_rates(grid, 0.1))
    f0 = np.array([0.25 + 0.03125 * i for i in range(grid.Q)])        # exact in single precision, all different
    for o in range(1, 2 * grid.dim + 1):
        want = f0.copy()
        for i, j in tables[name]['slip_swap_pairs'][str(o)]:
            want[i], want[j] = f0[j], f0[i]
        got, _, _ = oracle.node_update(desc, hipabi.SLF_NK_SLIP, o, None, f0, precision)
        assert np.array_equal(got, want), (o, got, want)
        # twice = nothing (a reflection), and every pair reverses the normal component only
        back, _, _ = oracle.node_update(desc, hipabi.SLF_NK_SLIP, o, None, got, precision)
        assert np.array_equal(back, f0)
        n = grid.basis[o]
        for i, j in tables[name]['slip_swap_pairs'][str(o)]:
            for d in range(grid.dim):
                assert grid.basis[j][d] == (-grid.basis[i][d] if n[d] else grid.basis[i][d])
    # no orientation: the reference's switch has no case for it
    same, _, _ = oracle.node_update(desc, hipabi.SLF_NK_SLIP, 0, None, f0, precision)
    assert np.array_equal(same, f0)


def _channel(grid, size, pattern, t_out, steps, precision='double', model='bgk'):
    dim = grid.dim
    params = [0.03, 0.0] + ([0.0] if dim == 3 else []) + [1.0]
    desc = make_box_desc(grid, size, model=model, precision=precision, access_pattern=pattern, visc=0.05, fluid_only=False,
                         type_kind=geo.TYPE_KIND_INPLACE, nt_bits=geo.NT_BITS, node_params=params,
                         periodic_fused=[0, 0, 1 if dim == 3 else 0])
    box = OracleBox(desc, periodic=(False, False, dim == 3),
                    node_map=geo.channel_inlet_outlet(desc, geo.T_ZHVEL, t_out, dim))
    shape = tuple(reversed(size))
    box.set_fields(np.ones(shape), [np.zeros(shape) for _ in range(dim)])
    box.initial_conditions()
    box.run(steps, save_last=True)
    return box


@pytest.mark.parametrize('grid,size', [(sym.D2Q9, (24, 12)), (sym.D3Q19, (14, 8, 5))])
@pytest.mark.parametrize('model', ['bgk', 'mrt'])
def test_do_nothing_outlet_in_place_is_the_two_copy_run(grid, size, model):
    """The reference's own criterion (tests/gpu/do_nothing_node.py: AA with NTDoNothing against AB), here to the bit:
    the node arithmetic is the same, and what the outlet reads for its unknown populations is the initial equilibrium
    in both (kept by the node in place; never overwritten in either copy of the two-copy run)."""
    for steps in (37, 60):
        ab = _channel(grid, size, 'AB', geo.T_DONOTHING, steps, model=model)
        aa = _channel(grid, size, 'AA', geo.T_DONOTHING, steps, model=model)
        for a, b in ((ab.rho, aa.rho),) + tuple(zip(ab.v[:grid.dim], aa.v[:grid.dim])):
            va, vb = ab.real_view(a), aa.real_view(b)
            mask = np.isfinite(va)
            assert np.array_equal(mask, np.isfinite(vb))
            assert np.array_equal(va[mask], vb[mask]), steps
    # ... and the flow has actually reached the outlet: the outlet column moves
    out_v = aa.real_view(aa.v[0])[..., 2:-2, -1]
    assert np.all(out_v > 1e-4)


@pytest.mark.parametrize('grid,size', [(sym.D2Q9, (24, 12)), (sym.D3Q19, (14, 8, 5))])
def test_without_the_node_type_the_in_place_outlet_is_something_else(grid, size):
    """What the node type is for: a plain fluid node at the outlet of an in-place run pulls its unknown populations
    from the ghost nodes behind it, which nobody has given a meaning (the oracle's arrays hold non-finite values there
    so that such a read shows) -- not the two-copy result."""
    ab = _channel(grid, size, 'AB', geo.T_FLUID, 60)
    aa = _channel(grid, size, 'AA', geo.T_FLUID, 60)
    va, vb = ab.real_view(ab.v[0]), aa.real_view(aa.v[0])
    wet = np.isfinite(ab.real_view(ab.rho))
    assert np.all(np.isfinite(va[wet]))
    assert not np.all(np.isfinite(vb[wet]))


@pytest.mark.parametrize('grid,size', [(sym.D2Q9, (24, 12)), (sym.D3Q19, (14, 8, 5))])
def test_do_nothing_node_keeps_its_unknown_populations(grid, size):
    """"The value from the previous time step is retained" (node_type.py:296-307): after any number of steps the
    populations an outlet node reads for its unknown directions are the ones it was initialised with."""
    dim = grid.dim
    o_out = 3 if dim == 2 else 2
    missing = [i for i in range(1, grid.Q) if sum(a * b for a, b in zip(grid.basis[i], grid.basis[o_out])) > 0]
    w = grid.weights_float
    for steps in (1, 2, 7, 20):
        box = _channel(grid, size, 'AA', geo.T_DONOTHING, steps)
        d = box.current_dist()
        nx, ny = size[0], size[1]
        for i in missing:
            opp = grid.idx_opposite[i]
            e = grid.basis[i]
            for y in range(2, ny):
                zs = [0] if dim == 2 else range(1, size[2] + 1)
                for z in zs:
                    if steps & 1:
                        # after an even-numbered step (the last one was iteration steps - 1 = even): the next step pulls
                        # from the opposite slot of the node behind
                        yy = y - e[1]
                        zz = z - (e[2] if dim == 3 else 0)
                        if dim == 3:
                            zz = (zz - 1) % size[2] + 1
                        got = d[opp, zz, yy, nx - e[0]]
                    else:
                        got = d[i, z, y, nx]
                    assert got == w[i], (steps, i, y, z, got, w[i])
