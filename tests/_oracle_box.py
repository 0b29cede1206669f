"""Oracle twin of sailfish_amd.box.BoxSim: the same step sequence executed by the
CPU oracle (oracle/lbm_oracle.c) on numpy arrays.  Test-only."""
import numpy as np

from oracle.oracle import OracleSim
from sailfish_amd import hipabi


class OracleBox(object):
    def __init__(self, desc, periodic=(False, False, False), node_map=None):
        self.desc = desc
        self.o = OracleSim(desc)
        self.dim, self.Q, self.dtype, self.shape = self.o.dim, self.o.Q, self.o.dtype, self.o.shape
        self.aa = desc.access_pattern == hipabi.SLF_AA
        self.pbc_axes = [a for a in range(self.dim) if periodic[a] and not desc.periodic_fused[a]]
        for a in range(self.dim):
            desc.periodic_local[a] = int(bool(periodic[a]))
        self.dist = [self.o.new_dist()]
        if not self.aa:
            self.dist.append(self.o.new_dist())
        self.rho = np.full(self.shape, np.inf, dtype=self.dtype)
        self.v = [np.full(self.shape, np.inf, dtype=self.dtype) for _ in range(3)]
        self.node_map = None if node_map is None else np.ascontiguousarray(node_map, dtype=np.uint32).reshape(self.shape)
        self.iteration = 0

    def real_view(self, arr):
        d = self.desc
        if self.dim == 3:
            return arr[..., 1:d.lat_nz - 1, 1:d.lat_ny - 1, 1:d.lat_nx - 1]
        return arr[..., 0, 1:d.lat_ny - 1, 1:d.lat_nx - 1]

    def set_fields(self, rho, v):
        self.real_view(self.rho)[...] = rho
        for d in range(self.dim):
            self.real_view(self.v[d])[...] = v[d]

    def initial_conditions(self):
        with np.errstate(all='ignore'):
            for d in self.dist:
                self.o.init(d, self.rho, self.v[0], self.v[1], self.v[2])
        self.iteration = 0

    def step(self, save_macro=False, region=None):
        it = self.iteration
        opts = 1 if save_macro else 0
        if self.aa:
            prop = 2 if (it & 1) else 1
            self.o.step(prop, self.node_map, self.dist[0], self.dist[0], self.rho, self.v[0], self.v[1], self.v[2],
                        opts, region)
            out, swap = 0, (it & 1) == 0
        else:
            i = it & 1
            self.o.step(0, self.node_map, self.dist[i], self.dist[1 - i], self.rho, self.v[0], self.v[1],
                        self.v[2], opts, region)
            out, swap = 1 - i, False
        for axis in self.pbc_axes:
            self.o.pbc(self.dist[out], axis, swap)
        self.iteration += 1

    def run(self, n, save_last=True):
        for i in range(n):
            self.step(save_macro=(save_last and i == n - 1))

    def current_dist(self):
        return self.dist[0] if self.aa else self.dist[self.iteration & 1]


def synthetic_fields(size, dim, seed=1234, dtype=np.float64):
    """Non-trivial smooth initial state: rho = 1 + 1e-3 U[0,1),
    u = 0.05 (sin 2 pi y/L, sin 2 pi z/L, sin 2 pi x/L)."""
    rng = np.random.RandomState(seed)
    shape = tuple(reversed(size))
    rho = 1.0 + 1e-3 * rng.rand(*shape)
    if dim == 3:
        nz, ny, nx = shape
        z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing='ij')
        v = [0.05 * np.sin(2 * np.pi * y / ny), 0.05 * np.sin(2 * np.pi * z / nz), 0.05 * np.sin(2 * np.pi * x / nx)]
    else:
        ny, nx = shape
        y, x = np.meshgrid(np.arange(ny), np.arange(nx), indexing='ij')
        v = [0.05 * np.sin(2 * np.pi * y / ny), 0.05 * np.sin(2 * np.pi * x / nx)]
    return rho.astype(dtype), [c.astype(dtype) for c in v]
