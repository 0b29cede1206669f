"""Pure-NumPy periodic BGK (np.roll streaming, float64): the textbook scheme, independent of lbm_oracle.c.

TEST INFRASTRUCTURE ONLY (see lbm_oracle.c): tests/test_oracle_streaming.py checks the oracle's streaming against it,
and bench.py's cpu_baseline prints its MLUPS next to the OpenMP twin's "for context" (the
reference has no CPU compute path; its backend_dummy is a no-op stub)."""
import time

import numpy as np


def run(grid, rho, v, visc, steps):
    """grid: a lattice class of sailfish_amd.sym; rho, v[d]: arrays over the nodes (z, y, x order).  Returns (f, rho, u)."""
    e = grid.basis_array
    w = grid.weights_float
    dim = grid.dim

    def feq(rho, v):
        usq = sum(c * c for c in v)
        out = []
        for i in range(grid.Q):
            eu = sum(e[i][d] * v[d] for d in range(dim))
            out.append(w[i] * rho * (1 + 3 * eu + 4.5 * eu * eu - 1.5 * usq))
        return np.array(out)

    f = feq(rho, v)
    omega = 1.0 / ((6.0 * visc + 1.0) / 2.0)          # tau = (6 visc + 1) / 2, reference sym.py:847-848
    for _ in range(steps):
        r = f.sum(axis=0)
        u = [sum(e[i][d] * f[i] for i in range(grid.Q)) / r for d in range(dim)]
        f = f + omega * (feq(r, u) - f)
        for i in range(grid.Q):
            # numpy axis order is (z, y, x)
            shift = tuple(int(e[i][d]) for d in reversed(range(dim)))
            f[i] = np.roll(f[i], shift, axis=tuple(range(dim)))
    r = f.sum(axis=0)
    u = [sum(e[i][d] * f[i] for i in range(grid.Q)) / r for d in range(dim)]
    return f, r, u


def mlups(grid, size, visc=1.0 / 6.0, steps=4):
    """MLUPS of the twin on a box of `size` nodes (x, y[, z])."""
    shape = tuple(reversed(size))
    rho = np.ones(shape)
    v = [np.zeros(shape) for _ in range(grid.dim)]
    run(grid, rho, v, visc, 1)
    t0 = time.perf_counter()
    run(grid, rho, v, visc, steps)
    return float(np.prod(size)) * steps / (time.perf_counter() - t0) * 1e-6
