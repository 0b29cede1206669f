"""Periodic box cut into z-slabs, one slab per GPU (the weak-scaling benchmark
geometry: SURVEY.md §8(d) M0/C4, reference geo.py:100-135 EqualSubdomainsGeometry3D
with conn_axis=z).

Per step (reference subdomain_runner.py:1028-1058 boundary/bulk split):
  calc stream : wait(previous halo) -> sweep plane z=1, sweep plane z=n -> event
                -> sweep interior planes
  halo stream : wait(event) -> pack the two face layers (index-list gather)
                -> RCCL send/recv with the two ring neighbours -> unpack -> event
x and y are wrapped inside the sweep; z is wrapped inside the sweep when there
is a single slab, otherwise it goes through the ghost planes + halo.

Which layer travels (reference subdomain_runner.py:1069-1103, Appendix A.4-5 of
SURVEY.md):
  push steps (AB, odd AA): my ghost plane (populations pushed across the face)
      -> neighbour's first real plane, same slots;
  even AA step (in-place, opposite slots): my first real plane, opposite slots
      -> neighbour's ghost plane, so that its next (odd) step can pull them.
"""
import numpy as np

from sailfish_amd import hipabi, sym
from sailfish_amd.box import BoxSim, make_box_desc


class SlabPlan(object):
    """Pure host logic: index lists (q * dist_size + gi, uint64) for the halo pack / unpack."""

    def __init__(self, grid, desc):
        self.grid, self.desc = grid, desc
        self.nx, self.ny, self.nz = desc.lat_nx - 2, desc.lat_ny - 2, desc.lat_nz - 2
        self.dist_size = hipabi.dist_stride(desc)
        self.up_dists = sym.get_prop_dists(grid, 1, 2)      # e_z = +1
        self.down_dists = sym.get_prop_dists(grid, -1, 2)   # e_z = -1
        self.count = len(self.up_dists) * self.nx * self.ny

    def _plane(self, z, dists):
        d = self.desc
        y, x = np.meshgrid(np.arange(1, self.ny + 1, dtype=np.uint64), np.arange(1, self.nx + 1, dtype=np.uint64),
                           indexing='ij')
        gi = x + np.uint64(d.arr_nx) * (y + np.uint64(d.arr_ny) * np.uint64(z))
        return np.concatenate([np.uint64(q) * np.uint64(self.dist_size) + gi.ravel() for q in dists])

    def lists(self, swap):
        """Returns (send_up, send_down, recv_low, recv_high) index arrays.
        swap = True after the even AA step."""
        n = self.nz
        opp = self.grid.idx_opposite
        if not swap:
            return (self._plane(n + 1, self.up_dists), self._plane(0, self.down_dists),
                    self._plane(1, self.up_dists), self._plane(n, self.down_dists))
        up_s = [opp[i] for i in self.up_dists]
        dn_s = [opp[i] for i in self.down_dists]
        return (self._plane(n, up_s), self._plane(1, dn_s), self._plane(0, up_s), self._plane(n + 1, dn_s))


class SlabSim(BoxSim):
    def __init__(self, backend, grid, size, rank=0, world=1, model='bgk', precision='single',
                 access_pattern='AA', visc=1.0 / 6.0, fused_periodic=True, exchanger=None):
        self.grid, self.size, self.rank, self.world = grid, size, rank, world
        assert grid.dim == 3
        if world > 1 and not fused_periodic:
            raise ValueError('multi-slab runs wrap x and y inside the sweep')
        fused = [int(fused_periodic), int(fused_periodic), int(fused_periodic and world == 1)]
        desc = make_box_desc(grid, size, model=model, precision=precision, access_pattern=access_pattern, visc=visc,
                             periodic_fused=fused, fluid_only=True)
        periodic = (True, True, world == 1)
        BoxSim.__init__(self, backend, desc, periodic=periodic)
        self.calc_stream = self.stream
        self.block_size = self.module.block_size
        if world > 1:
            self._init_halo(exchanger)

    # -- halo machinery ------------------------------------------------------
    def _init_halo(self, exchanger):
        import torch
        from sailfish_amd.connector import RingExchanger, init_distributed
        init_distributed()
        b = self.backend
        self.plan = SlabPlan(self.grid, self.desc)
        self.exchanger = exchanger or RingExchanger(self.rank, self.world)
        self.halo_stream = b.make_stream()
        self.t_halo_stream = torch.cuda.ExternalStream(self.halo_stream.native, device=torch.device('cuda', b.gpu_id))
        tdtype = torch.float32 if self.desc.precision == 4 else torch.float64
        n = self.plan.count
        dev = torch.device('cuda', b.gpu_id)
        self.t_bufs = [torch.empty(n, dtype=tdtype, device=dev) for _ in range(4)]  # s_up s_down r_low r_high
        self._idx_keep = []
        self.k_halo = {}
        for swap in ((False, True) if self.aa else (False,)):
            lists = self.plan.lists(swap)
            gpu_idx = []
            for a in lists:
                a = np.ascontiguousarray(a, dtype=np.uint64)
                self._idx_keep.append(a)
                gpu_idx.append(b.alloc_buf(like=a))
            for di, dbuf in enumerate(self.gpu_dist):
                ks = []
                for j in range(4):
                    name = 'CollectSparseData' if j < 2 else 'DistributeSparseData'
                    ks.append(b.get_kernel(self.module, name, (64,),
                                           [gpu_idx[j], dbuf, self.t_bufs[j].data_ptr(), n], 'PPPi'))
                self.k_halo[(swap, di)] = ks
        self.ev_halo = None
        nz = self.desc.lat_nz - 2
        ny = self.desc.lat_ny - 2
        self.reg_low, self.reg_high = (1, ny + 1, 1, 2), (1, ny + 1, nz, nz + 1)
        self.reg_bulk = (1, ny + 1, 2, nz)

    def step_compute(self, save_macro=False):
        """Boundary planes, event, bulk planes on the calc stream; halo pack on the halo stream."""
        b = self.backend
        it = self.iteration
        if self.aa:
            k, out, swap = self.k_sweep[int(save_macro)][0], 0, (it & 1) == 0
        else:
            k, out, swap = self.k_sweep[int(save_macro)][it & 1], 1 - (it & 1), False
        if self.ev_halo is not None:
            self.calc_stream.wait_for_event(self.ev_halo)
        b.run_kernel(k, self.reg_low, self.calc_stream)
        b.run_kernel(k, self.reg_high, self.calc_stream)
        ev_bnd = b.make_event(self.calc_stream)
        b.run_kernel(k, self.reg_bulk, self.calc_stream)
        self.halo_stream.wait_for_event(ev_bnd)
        self._ks = self.k_halo[(swap, out)]
        b.run_kernel(self._ks[0], None, self.halo_stream)
        b.run_kernel(self._ks[1], None, self.halo_stream)
        self.iteration += 1
        b.set_iteration(self.iteration)

    def step_exchange(self):
        import torch
        with torch.cuda.stream(self.t_halo_stream):
            self.exchanger.exchange(*self.t_bufs)

    def step_finish(self):
        b = self.backend
        b.run_kernel(self._ks[2], None, self.halo_stream)
        b.run_kernel(self._ks[3], None, self.halo_stream)
        self.ev_halo = b.make_event(self.halo_stream)

    def step(self, save_macro=False, region=None):
        if self.world == 1:
            return BoxSim.step(self, save_macro)
        self.step_compute(save_macro)
        self.step_exchange()
        self.step_finish()

    def sync(self):
        self.stream.synchronize()
        if self.world > 1:
            self.halo_stream.synchronize()

    # -- initial state ---------------------------------------------------------
    def init_synthetic(self, seed=1234):
        """rho = 1 + 1e-3 U[0,1), u = 0.05 (sin 2 pi y/Ly, sin 2 pi z/Lz, sin 2 pi x/Lx),
        z measured in the global (all slabs) box."""
        nx, ny, nz = self.size
        rng = np.random.RandomState(seed + self.rank)
        rho = (1.0 + 1e-3 * rng.rand(nz, ny, nx)).astype(self.dtype)
        x = np.arange(nx, dtype=np.float64)
        y = np.arange(ny, dtype=np.float64)
        z = np.arange(nz, dtype=np.float64) + self.rank * nz
        vx = np.broadcast_to((0.05 * np.sin(2 * np.pi * y / ny))[None, :, None], (nz, ny, nx))
        vy = np.broadcast_to((0.05 * np.sin(2 * np.pi * z / (nz * self.world)))[:, None, None], (nz, ny, nx))
        vz = np.broadcast_to((0.05 * np.sin(2 * np.pi * x / nx))[None, None, :], (nz, ny, nx))
        self.set_fields(rho, [vx.astype(self.dtype), vy.astype(self.dtype), vz.astype(self.dtype)])
        self.initial_conditions()
        self.sync()
