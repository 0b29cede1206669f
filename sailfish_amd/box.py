"""Single-subdomain driver on top of the backend interface.

A thin, allocation-owning helper used by bench.py, __graft_entry__.smoke() and
the GPU parity tests: it performs exactly the sequence of backend calls the
subdomain runner's step() performs for one subdomain without neighbours
(reference sailfish/subdomain_runner.py:960-1009: CollideAndPropagate, then
ApplyPeriodicBoundaryConditions per periodic axis, A/B swap or AA parity), but
takes plain numpy inputs instead of the LBSim/Subdomain classes.
"""
import os

import numpy as np

from sailfish_amd import hipabi, placement, sym


def padded_nx(lat_nx, alignment=32):
    """x padding, reference subdomain_runner.py:367-373 (--mem_alignment, default 32 nodes)."""
    return int(np.ceil(float(lat_nx) / alignment)) * alignment


def make_box_desc(grid, size, model='bgk', precision='single', access_pattern='AA', visc=1.0 / 6.0,
                  periodic_fused=(0, 0, 0), fluid_only=True, accel=None, incompressible=False,
                  relaxation_enabled=True, type_kind=None, node_params=None, nt_bits=None, use_link_tags=True,
                  alignment=32, dist_pad=None, regularized=False, subgrid=None, smagorinsky_const=0.1):
    """size = (nx, ny[, nz]) real nodes; a ghost envelope of 1 is added."""
    dim = grid.dim
    assert len(size) == dim
    lat = [s + 2 for s in size]
    kw = dict(lattice=grid.slf_id,
              model=hipabi.SLF_MRT if model == 'mrt' else hipabi.SLF_BGK,
              precision=4 if precision == 'single' else 8,
              access_pattern=hipabi.SLF_AA if access_pattern == 'AA' else hipabi.SLF_AB,
              lat_nx=lat[0], lat_ny=lat[1], lat_nz=lat[2] if dim == 3 else 1,
              arr_nx=padded_nx(lat[0], alignment),
              arr_ny=lat[1], arr_nz=lat[2] if dim == 3 else 1,
              periodic_fused=list(periodic_fused), fluid_only=int(fluid_only),
              tau=sym.relaxation_time(visc), visc=visc, mrt_rates=sym.mrt_rates(grid, visc),
              incompressible=int(incompressible), relaxation_enabled=int(relaxation_enabled),
              use_link_tags=int(use_link_tags))
    if accel is not None:
        kw['has_force'] = 1
        kw['accel'] = list(accel) + [0.0] * (3 - len(accel))
    if type_kind is not None:
        kw['type_kind'] = type_kind
    if node_params is not None:
        kw['node_params'] = node_params
    if nt_bits is not None:
        misc, param, scratch = nt_bits
        kw.update(nt_type_mask=(1 << misc) - 1, nt_misc_shift=misc, nt_param_shift=param, nt_scratch_shift=scratch)
    if dist_pad:
        kw['dist_stride'] = kw['arr_nx'] * kw['arr_ny'] * kw['arr_nz'] + int(dist_pad)
    if regularized or subgrid:
        kw.update(regularized=int(bool(regularized)), smagorinsky_const=float(smagorinsky_const),
                  subgrid=hipabi.SLF_SUBGRID_LES_SMAGORINSKY if subgrid else hipabi.SLF_SUBGRID_NONE)
    return hipabi.make_desc(**kw)


class BoxSim(object):
    """One subdomain, no neighbours, on one GPU through the backend interface."""

    def __init__(self, backend, desc, periodic=(False, False, False), node_map=None, tune_placement=False):
        self.backend = backend
        self.desc = desc
        self.dim = 2 if desc.lattice == hipabi.SLF_D2Q9 else 3
        self.Q = 9 if self.dim == 2 else 19
        self.dtype = np.float32 if desc.precision == 4 else np.float64
        self.shape = (desc.arr_nz, desc.arr_ny, desc.arr_nx)
        self.nodes = desc.arr_nz * desc.arr_ny * desc.arr_nx
        self.aa = desc.access_pattern == hipabi.SLF_AA
        # axes that need ghost-layer PBC kernels = periodic and not wrapped by the sweep
        self.pbc_axes = [a for a in range(self.dim) if periodic[a] and not desc.periodic_fused[a]]
        self.iteration = 0
        for a in range(self.dim):
            desc.periodic_local[a] = int(bool(periodic[a]))
        b = backend
        self.stride = hipabi.dist_stride(desc)
        fbytes = self.stride * self.dtype().itemsize
        self.module = b.build(desc)
        off = b.dist_align_offset(self.dtype().itemsize)
        # large distribution arrays are *placed*: spread over physical HBM (placement.py)
        self.placed = []
        self.placement_info = None
        self.placement_tuning = None
        if placement.enabled() and self.Q * fbytes >= placement.MIN_BYTES and not int(desc.node_addressing):
            sizes = [self.Q * fbytes] * (1 if self.aa else 2)
            if tune_placement and os.environ.get('SLF_PLACEMENT_TUNE', '1') != '0' and not placement.holding_now():
                # placement by measurement (placement.choose): what SubdomainRunner does for its arrays
                probe_stream = b.make_stream()
                self.placed, self.placement_tuning = placement.choose(
                    lambda: b.alloc_placed(sizes, off),
                    lambda bs: placement.probe_sweep(b, desc, self.dim, bs[0].addr, bs[-1].addr, self.Q * fbytes, probe_stream),
                    lambda bs: [b.free_buf(pb.addr) for pb in bs], room=lambda: placement.room_for(b, sum(sizes)))
            else:
                self.placed = b.alloc_placed(sizes, off)
            self.placement_info = b.last_placement
            self.gpu_dist = [pb.addr for pb in self.placed]
        else:
            self.gpu_dist = [b.alloc_buf(size=self.Q * fbytes, align_offset=off)]
            if not self.aa:
                self.gpu_dist.append(b.alloc_buf(size=self.Q * fbytes, align_offset=off))
        # host mirrors of the macroscopic fields (ghost = +inf sentinel, reference
        # subdomain_runner.py:278-297)
        self.rho = np.full(self.shape, np.inf, dtype=self.dtype)
        self.v = [np.full(self.shape, np.inf, dtype=self.dtype) for _ in range(self.dim)]
        foff = b.dist_align_offset(self.dtype().itemsize)      # x = 1 of every field row on a 128-byte line as well
        self.gpu_rho = b.alloc_buf(like=self.rho, align_offset=foff)
        self.gpu_v = [b.alloc_buf(like=a, align_offset=foff) for a in self.v]
        self.gpu_map = 0
        self.node_map = None
        if node_map is not None:
            self.node_map = np.ascontiguousarray(node_map, dtype=np.uint32).reshape(self.shape)
            self.gpu_map = b.alloc_buf(like=self.node_map, align_offset=b.dist_align_offset(4))
        self.stream = b.make_stream()
        self.row_classes = None
        if self.gpu_map and b.supports_row_classes(desc) and os.environ.get('SLF_ROW_CLASSES', '1') != '0':
            self.row_classes = b.classify_rows(self.module, self.gpu_map, self.stream)
        self._make_kernels()

    def real_view(self, arr):
        d = self.desc
        if self.dim == 3:
            return arr[..., 1:d.lat_nz - 1, 1:d.lat_ny - 1, 1:d.lat_nx - 1]
        return arr[..., 0, 1:d.lat_ny - 1, 1:d.lat_nx - 1]

    def _make_kernels(self):
        b, m = self.backend, self.module
        sig = 'P' * (4 + self.dim) + 'i'
        self.k_sweep = {}
        for save in (0, 1):
            ks = []
            pairs = [(0, 0)] if self.aa else [(0, 1), (1, 0)]
            for i, o in pairs:
                args = [self.gpu_map, self.gpu_dist[i], self.gpu_dist[o], self.gpu_rho] + self.gpu_v + [save]
                ks.append(b.get_kernel(m, 'CollideAndPropagate', (64,), args, sig, needs_iteration=self.aa))
            self.k_sweep[save] = ks
        self.k_init = []
        for dbuf in self.gpu_dist:
            args = [dbuf] + self.gpu_v + [self.gpu_rho, self.gpu_map]
            self.k_init.append(b.get_kernel(m, 'SetInitialConditions', (64,), args, 'P' * (3 + self.dim)))
        self.k_pbc = {}
        for i, dbuf in enumerate(self.gpu_dist):
            for axis in self.pbc_axes:
                self.k_pbc[(i, axis, False)] = b.get_kernel(m, 'ApplyPeriodicBoundaryConditions', (64,),
                                                            [dbuf, axis], 'Pi')
                if self.aa:
                    self.k_pbc[(i, axis, True)] = b.get_kernel(m, 'ApplyPeriodicBoundaryConditionsWithSwap',
                                                               (64,), [dbuf, axis], 'Pi')
        args = [self.gpu_map, self.gpu_dist[0], self.gpu_dist[0], self.gpu_rho] + self.gpu_v + [1]
        self.k_macro = b.get_kernel(m, 'ComputeMacroFields', (64,), args, sig, needs_iteration=self.aa)

    def set_fields(self, rho, v):
        """rho, v[d]: arrays over the *real* nodes ((nz,) ny, nx)."""
        self.real_view(self.rho)[...] = rho
        for d in range(self.dim):
            self.real_view(self.v[d])[...] = v[d]
        self.backend.to_buf(self.gpu_rho)
        for g in self.gpu_v:
            self.backend.to_buf(g)

    def initial_conditions(self):
        """SetInitialConditions on every dist copy (reference lb_single.py:72-94)."""
        for k in self.k_init:
            self.backend.run_kernel(k, None, self.stream)
        self.iteration = 0
        self.backend.set_iteration(0)

    def step(self, save_macro=False, region=None):
        b = self.backend
        it = self.iteration
        if self.aa:
            b.run_kernel(self.k_sweep[int(save_macro)][0], region, self.stream)
            out, swap = 0, (it & 1) == 0
        else:
            b.run_kernel(self.k_sweep[int(save_macro)][it & 1], region, self.stream)
            out, swap = 1 - (it & 1), False
        for axis in self.pbc_axes:
            b.run_kernel(self.k_pbc[(out, axis, swap)], None, self.stream)
        self.iteration += 1
        b.set_iteration(self.iteration)

    def run(self, n, save_last=True):
        for i in range(n):
            self.step(save_macro=(save_last and i == n - 1))

    def sync(self):
        self.stream.synchronize()

    def release(self):
        """Frees the device memory of this simulation."""
        self.sync()
        b = self.backend
        for addr in list(self.gpu_dist) + [self.gpu_rho] + list(self.gpu_v) + ([self.gpu_map] if self.gpu_map else []):
            b.free_buf(addr)
        self.gpu_dist = []
        # drop the kernel objects: the backend's registry of iteration-dependent kernels holds them weakly
        for name in ('k_init', 'k_sweep', 'k_pbc', 'k_macro', 'k_halo'):
            if hasattr(self, name):
                setattr(self, name, None)

    def fetch_fields(self):
        self.sync()
        self.backend.from_buf(self.gpu_rho)
        for g in self.gpu_v:
            self.backend.from_buf(g)
        return self.rho, self.v

    def current_dist_index(self):
        return 0 if self.aa else (self.iteration & 1)

    def get_dist(self, which=None):
        """Raw distributions [Q, (nz,) ny, arr_nx] (reference _debug_get_dist, subdomain_runner.py:1363-1381)."""
        self.sync()
        idx = self.current_dist_index() if which is None else which
        raw = np.zeros((self.Q, self.stride), dtype=self.dtype)
        self.backend.from_buf(self.gpu_dist[idx], raw)
        return np.ascontiguousarray(raw[:, :self.nodes]).reshape((self.Q,) + self.shape)

    def set_dist(self, host, which=None):
        idx = self.current_dist_index() if which is None else which
        raw = np.zeros((self.Q, self.stride), dtype=self.dtype)
        raw[:, :self.nodes] = np.asarray(host, dtype=self.dtype).reshape(self.Q, self.nodes)
        self.backend.to_buf(self.gpu_dist[idx], raw)
