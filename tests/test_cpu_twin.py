"""The blocked OpenMP CPU twin (oracle/lbm_fast.c, the cpu_baseline of bench.py) must agree with the table-driven
oracle bit for bit: populations and macroscopic fields, D3Q19 and D2Q9, even and odd step counts, several threads."""
import numpy as np
import pytest

from oracle import cpu_twin
from sailfish_amd import sym
from sailfish_amd.box import make_box_desc
from tests._oracle_box import OracleBox, synthetic_fields


@pytest.mark.parametrize('grid,size', [(sym.D3Q19, (37, 9, 7)), (sym.D3Q19, (64, 5, 4)), (sym.D2Q9, (45, 13)),
                                       (sym.D3Q19, (2, 3, 3))])
@pytest.mark.parametrize('precision', ['single', 'double'])
@pytest.mark.parametrize('steps', [1, 6, 7])
def test_twin_equals_oracle(grid, size, precision, steps):
    visc = 0.013
    desc = make_box_desc(grid, size, precision=precision, access_pattern='AA', visc=visc,
                         periodic_fused=[1] * 3)
    o = OracleBox(desc, periodic=(True, True, True))
    rho, v = synthetic_fields(size, grid.dim)
    o.set_fields(rho, v)
    o.initial_conditions()
    t = cpu_twin.FastBox('D3Q19' if grid.dim == 3 else 'D2Q9', size, visc, precision)
    assert t.shape == o.shape and t.arr_nx == desc.arr_nx
    t.set_dist(o.dist[0])
    o.run(steps, save_last=True)
    fields = [np.full(o.shape, np.inf, dtype=o.dtype) for _ in range(4)]
    t.run(steps, fields)
    assert np.array_equal(o.real_view(o.dist[0]), o.real_view(t.dist))
    assert np.array_equal(o.real_view(o.rho), o.real_view(fields[0]))
    for d in range(grid.dim):
        assert np.array_equal(o.real_view(o.v[d]), o.real_view(fields[1 + d]))


def test_baseline_object_shape():
    b = cpu_twin.baseline(budget_s=0.5) if False else None   # timing run: exercised by bench.py on the GPU box
    box = cpu_twin.FastBox('D3Q19', (32, 32, 32), 1.0 / 6.0)
    box.init_uniform()
    assert box.mlups(2, repeats=1) > 0
    m0 = box.dist.astype(np.float64)[:, 1:-1, 1:-1, 1:33].sum()
    box.run(10)
    m1 = box.dist.astype(np.float64)[:, 1:-1, 1:-1, 1:33].sum()
    assert abs(m1 - m0) / m0 < 1e-6 and b is None
