"""CPU-only checks of the oracle's streaming / periodic-boundary logic:
  * against an independent pure-numpy (np.roll) periodic LBM twin,
  * AB == AA after an even number of steps (the invariant the reference checks
    on GPU in tests/gpu/access_pattern.sh:12-29),
  * ghost-layer PBC kernels == in-sweep periodic wrap,
  * mass / momentum conservation.
"""
import numpy as np
import pytest

from sailfish_amd import sym
from sailfish_amd.box import make_box_desc
from tests._oracle_box import OracleBox, synthetic_fields


from oracle.numpy_twin import run as numpy_twin  # noqa: E402  (the textbook np.roll scheme)


CASES = [(sym.D2Q9, (12, 9)), (sym.D3Q19, (9, 7, 6))]


@pytest.mark.parametrize('grid,size', CASES)
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
@pytest.mark.parametrize('fused', [False, True])
def test_oracle_vs_numpy_twin(grid, size, pattern, fused):
    steps = 6
    rho, v = synthetic_fields(size, grid.dim)
    desc = make_box_desc(grid, size, precision='double', access_pattern=pattern, visc=0.05,
                         periodic_fused=[int(fused)] * 3)
    ob = OracleBox(desc, periodic=(True, True, True))
    ob.set_fields(rho, v)
    ob.initial_conditions()
    ob.run(steps, save_last=False)
    f_ref, r_ref, u_ref = numpy_twin(grid, rho, v, 0.05, steps)
    got = ob.real_view(ob.current_dist())
    assert np.max(np.abs(got - f_ref)) < 1e-13
    # macroscopic fields written by the *next* step are those of the state after `steps`
    ob.step(save_macro=True)
    assert np.max(np.abs(ob.real_view(ob.rho) - r_ref)) < 1e-13
    for d in range(grid.dim):
        assert np.max(np.abs(ob.real_view(ob.v[d]) - u_ref[d])) < 1e-13


@pytest.mark.parametrize('grid,size', CASES)
@pytest.mark.parametrize('model', ['bgk', 'mrt'])
@pytest.mark.parametrize('precision', ['single', 'double'])
def test_oracle_ab_equals_aa_bit_exact(grid, size, model, precision):
    rho, v = synthetic_fields(size, grid.dim)
    res = {}
    for pattern in ('AB', 'AA'):
        desc = make_box_desc(grid, size, model=model, precision=precision, access_pattern=pattern, visc=0.02)
        ob = OracleBox(desc, periodic=(True, True, True))
        ob.set_fields(rho, v)
        ob.initial_conditions()
        ob.run(8, save_last=True)
        res[pattern] = (ob.real_view(ob.current_dist()).copy(), ob.real_view(ob.rho).copy())
    assert np.array_equal(res['AB'][0], res['AA'][0])
    assert np.array_equal(res['AB'][1], res['AA'][1])


@pytest.mark.parametrize('grid,size', CASES)
@pytest.mark.parametrize('pattern', ['AB', 'AA'])
def test_oracle_ghost_pbc_equals_fused_wrap(grid, size, pattern):
    rho, v = synthetic_fields(size, grid.dim)
    res = []
    for fused in (0, 1):
        desc = make_box_desc(grid, size, precision='single', access_pattern=pattern, visc=0.01,
                             periodic_fused=[fused] * 3)
        ob = OracleBox(desc, periodic=(True, True, True))
        ob.set_fields(rho, v)
        ob.initial_conditions()
        ob.run(7, save_last=False)   # odd count: AA is in its swapped state, compare after one more
        ob.step(save_macro=True)
        res.append((ob.real_view(ob.current_dist()).copy(), ob.real_view(ob.rho).copy()))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize('grid,size', CASES)
def test_oracle_partial_periodicity(grid, size):
    """Only some axes periodic: ghost-PBC and fused wrap must still agree on every real node
    that cannot be reached from a non-periodic face within the run."""
    rho, v = synthetic_fields(size, grid.dim)
    per = [True] + [False] * (grid.dim - 1)
    res = []
    for fused in (0, 1):
        desc = make_box_desc(grid, size, precision='double', access_pattern='AB', visc=0.01,
                             periodic_fused=[fused if p else 0 for p in per] + [0] * (3 - grid.dim))
        ob = OracleBox(desc, periodic=per + [False] * (3 - grid.dim))
        ob.set_fields(rho, v)
        ob.initial_conditions()
        ob.run(1, save_last=False)
        res.append(ob.real_view(ob.current_dist()).copy())
    inner = (slice(None),) + tuple(slice(1, -1) for _ in range(grid.dim - 1)) + (slice(None),)
    assert np.array_equal(res[0][inner], res[1][inner])


@pytest.mark.parametrize('grid,size', CASES)
def test_oracle_conservation(grid, size):
    rho, v = synthetic_fields(size, grid.dim)
    desc = make_box_desc(grid, size, model='mrt', precision='double', access_pattern='AA', visc=0.03,
                         periodic_fused=[1, 1, 1])
    ob = OracleBox(desc, periodic=(True, True, True))
    ob.set_fields(rho, v)
    ob.initial_conditions()
    f0 = ob.real_view(ob.current_dist()).copy()
    ob.run(10, save_last=False)
    f1 = ob.real_view(ob.current_dist())
    assert abs(f0.sum() - f1.sum()) < 1e-10
    e = grid.basis_array
    for d in range(grid.dim):
        m0 = sum(e[i][d] * f0[i].sum() for i in range(grid.Q))
        m1 = sum(e[i][d] * f1[i].sum() for i in range(grid.Q))
        assert abs(m0 - m1) < 1e-10
