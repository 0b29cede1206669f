"""Per-subdomain driver: memory, kernels, the time-step loop, halo exchange,
output and checkpoints (reference sailfish/subdomain_runner.py).

One runner = one subdomain on one GPU.  The step keeps the reference's structure
(subdomain_runner.py:960-1139):

    calc stream: [wait for the previous halo]  sweep(boundary regions) -> event -> sweep(bulk)
                 -> periodic-boundary kernels of locally periodic axes
    halo stream: wait(event) -> pack (index-list gather) -> exchange -> unpack -> event

but everything stays on the device: the pack kernels write straight into the
send buffer, buffers travel GPU-to-GPU (RCCL over xGMI through torch.distributed,
or a peer copy when both subdomains live in this process), the unpack kernels
scatter into the neighbour-owned slots.  The reference stages every halo through
pinned host memory and a zmq socket (subdomain_runner.py:1064-1139).

Launch geometry is a region of rows / planes (`backend.run_kernel(kernel, region)`)
instead of the reference's bulk / boundary block arithmetic (:396-475).
"""
import math
import os
import pickle
import time

import numpy as np

from sailfish_amd import hipabi, io, subdomain_connection, util, xface
from sailfish_amd import node_type as nt
from sailfish_amd.lb_base import LBSim  # noqa: F401  (type reference)
from sailfish_amd.profile import TimeProfile
from sailfish_amd.stepqueue import DirectQueue, NotPlannable


class GPUBuffer(object):
    """Host/device buffer pair (reference subdomain_runner.py:29-42)."""

    def __init__(self, host_buffer, backend):
        self.host = host_buffer
        self.gpu = backend.alloc_buf(like=host_buffer) if host_buffer is not None else None


class SubdomainRunner(object):
    """Runs the simulation for a single SubdomainSpec."""

    def __init__(self, simulation, spec, output, backend, quit_event=None, summary_addr=None, master_addr=None,
                 summary_channel=None):
        self._sim = simulation
        self._spec = spec
        self._output = output
        self.backend = backend
        self._quit_event = quit_event
        self.config = simulation.config
        self._spec.runner = self
        self._bcg = None
        self._scalar_fields = []
        self._vector_fields = []
        self._gpu_field_map = {}
        self._host_base = {}
        self._gpu_grids_primary = []
        self._gpu_grids_secondary = []
        self._kernels_prepared = False
        self._links = {}
        self._connector = None
        self._all_specs = None
        self._global_size = None
        self._global_periodic = None
        self.timing = {'steps': 0, 'wall': 0.0}
        self._profile = TimeProfile(self)
        self._checkpoint_req = False      # set by SIGHUP (reference sighup_handler, :1528-1535)
        self._init_geometry_done = False
        if not hasattr(self.config, 'logger'):
            self.config.logger = util.setup_logger(self.config)

    # ------------------------------------------------------------------ geometry
    @property
    def dim(self):
        return self._spec.dim

    def set_topology(self, all_specs, global_size, periodic):
        """all_specs: every SubdomainSpec of the simulation (for halo routing)."""
        self._all_specs = all_specs
        self._global_size = list(global_size)
        self._global_periodic = list(periodic)

    def ghost_owner_map(self, spec):
        if self._all_specs is None or len(self._all_specs) < 2:
            return np.zeros(list(reversed(spec.actual_size)), dtype=bool)
        return subdomain_connection.ghost_owned_by_others(spec, self._all_specs, self._global_size,
                                                          self._global_periodic)

    def _init_shape(self):
        """Logical (ghost-including) and physical (x padded) sizes, reference subdomain_runner.py:359-373."""
        self._lat_size = list(reversed(self._spec.actual_size))
        self._physical_size = list(reversed(self._spec.actual_size))
        alignment = self.config.mem_alignment
        self._physical_size[-1] = int(math.ceil(float(self._physical_size[-1]) / alignment)) * alignment
        if self._global_size is None:
            gs = [self.config.lat_nx, self.config.lat_ny] + ([self.config.lat_nz] if self.dim == 3 else [])
            per = [self.config.periodic_x, self.config.periodic_y] + \
                ([self.config.periodic_z] if self.dim == 3 else [])
            self.set_topology([self._spec], gs, per)
        self._global_shape = tuple(reversed(self._global_size))

    def _get_nodes(self):
        return int(np.prod(self._physical_size))

    @property
    def num_phys_nodes(self):
        return self._get_nodes()

    @property
    def float(self):
        return np.float32 if self.config.precision == 'single' else np.float64

    def _lat_view(self, buf):
        sl = tuple(slice(0, n) for n in self._lat_size)
        return buf[sl]

    def make_scalar_field(self, dtype=None, name=None, register=True, async_=False, gpu_array=False,
                          nonghost_view=True):
        """Allocates a host array of the padded size and returns a view of the lattice (ghost nodes
        excluded unless nonghost_view=False).  Float fields are filled with +inf in the ghost layer and 0
        elsewhere (reference subdomain_runner.py:253-320)."""
        if dtype is None:
            dtype = self.float
        size = self._get_nodes()
        if async_:
            buf = self.backend.alloc_async_host_buf(size, dtype)
        else:
            buf = np.zeros(size, dtype=dtype)
        buf = buf.reshape(self._physical_size)
        if np.issubdtype(dtype, np.floating):
            buf[:] = np.inf
        fview = self._lat_view(buf)
        self._host_base[id(fview)] = buf
        if nonghost_view:
            view = fview[self._spec._nonghost_slice]
            if np.issubdtype(dtype, np.floating):
                view[:] = 0.0
            self._host_base[id(view)] = buf
        else:
            view = fview
        self._lat_views = getattr(self, '_lat_views', {})
        self._lat_views[id(view)] = buf
        if register:
            self._scalar_fields.append(view)
            if name is not None and self._output is not None:
                self._output.register_field(view, name)
        return view, None

    def make_vector_field(self, name=None, output=False, async_=False, gpu_array=False):
        components = [self.make_scalar_field(self.float, register=False, async_=async_)[0]
                      for _ in range(self.dim)]
        self._vector_fields.append(components)
        if name is not None and self._output is not None:
            self._output.register_field(components, name)
        return components

    def field_base(self, field):
        """The whole padded host array (ghosts + x padding) behind a field returned by make_scalar_field."""
        return self._lat_views[id(field)]

    def visualization_map(self):
        return self._subdomain.visualization_map()

    def _init_geometry(self):
        self._init_shape()
        self._subdomain = self._sim.subdomain(self._global_shape, self._spec, self._sim.grid)
        self._subdomain.allocate()
        self._subdomain.reset()
        self._init_geometry_done = True

    # ------------------------------------------------------------------ compute setup
    def _local_periodic(self):
        return [bool(self._spec._periodicity[a]) for a in range(self.dim)]

    def _module_desc(self):
        cfg = self.config
        lat = list(reversed(self._lat_size))
        arr = list(reversed(self._physical_size))
        kw = dict(precision=4 if cfg.precision == 'single' else 8,
                  access_pattern=hipabi.SLF_AA if cfg.access_pattern == 'AA' else hipabi.SLF_AB,
                  lat_nx=lat[0], lat_ny=lat[1], lat_nz=lat[2] if self.dim == 3 else 1,
                  arr_nx=arr[0], arr_ny=arr[1], arr_nz=arr[2] if self.dim == 3 else 1)
        self._sim.fill_module_desc(kw)
        kw.update(self._subdomain._encoder.desc_fields())
        if hasattr(self._sim, 'check_module_desc'):
            self._sim.check_module_desc(kw)
        local = self._local_periodic()
        fused = [int(local[a] and getattr(cfg, 'hip_fused_periodic', True)) for a in range(self.dim)]
        kw['periodic_local'] = [int(x) for x in local] + [0] * (3 - self.dim)
        kw['periodic_fused'] = fused + [0] * (3 - self.dim)
        self._fused = fused
        # fast path: every real node is a plain fluid node -> the sweep does not read the node map (ghost
        # nodes behind an unconnected, non-periodic face are never pulled from by a fluid-only interior
        # in any way that matters: what they hold is what a wall of excluded nodes would hold)
        vis = self._subdomain.visualization_map()
        kw['fluid_only'] = int(np.all(vis == 0))
        excluded = np.isin(vis, [nt._NTUnused.id, nt._NTGhost.id, nt._NTPropagationOnly.id])
        kw['sparse_geometry'] = int(excluded.mean() > 0.05)
        if self.indirect:
            # distributions hold the active nodes only; everything else stays dense
            if any(local[a] and not fused[a] for a in range(self.dim)):
                raise ValueError('--node_addressing=indirect needs in-sweep periodic boundaries '
                                 '(do not combine with --nohip_fused_periodic)')
            kw['fluid_only'] = 0
            kw['node_addressing'] = hipabi.SLF_ADDR_INDIRECT
            kw['dist_stride'] = (self._subdomain.active_nodes + 1 + 31) // 32 * 32   # + one spare slot (halo lists)
        return hipabi.make_desc(**kw)

    @property
    def indirect(self):
        return getattr(self.config, 'node_addressing', 'direct') == 'indirect'

    def _indirect_address_host(self):
        """Dense table node -> slot (or SLF_INVALID_NODE) over the padded arrays, slots numbered in memory
        order of the active nodes so that neighbouring active nodes have neighbouring slots
        (reference _build_indirect_address_map, :829-837)."""
        addr, _ = self.make_scalar_field(np.uint32, register=False, nonghost_view=False)
        base = self._host_base[id(addr)]
        base[:] = hipabi.SLF_INVALID_NODE
        mask = self._subdomain.active_node_mask
        addr[mask] = np.arange(int(mask.sum()), dtype=np.uint32)
        return base

    def _build_indirect_address_map(self):
        self._host_indirect_address = self._indirect_address_host()
        self._gpu_indirect_address = self.backend.alloc_buf(like=self._host_indirect_address)

    def _translate_halo_links(self, links, addr_base, dense_nodes, dist_stride):
        """Halo index lists (population * dense_nodes + dense node) -> (population * dist_stride + slot).
        Halo nodes next to solid regions own no slot: they are routed to the spare slot behind the last
        active node, so that both sides of a link keep lists of the same length and order."""
        addr = addr_base.reshape(-1).astype(np.uint64)
        spare = np.uint64(self._subdomain.active_nodes)
        for link in links.values():
            for kind in ('push_send', 'push_recv', 'pull_send', 'pull_recv'):
                idx = getattr(link, kind)
                q, node = idx // np.uint64(dense_nodes), idx % np.uint64(dense_nodes)
                slot = addr[node.astype(np.int64)]
                slot[slot == np.uint64(hipabi.SLF_INVALID_NODE)] = spare
                setattr(link, kind, np.ascontiguousarray(q * np.uint64(dist_stride) + slot, dtype=np.uint64))
        return links

    def gpu_indirect_address(self):
        return self._gpu_indirect_address

    def add_indirect_args(self, args, signature):
        """Indirect modules take the address table as an additional first kernel argument
        (reference _add_indirect_args, :1153-1157)."""
        if self.indirect:
            return [self.gpu_indirect_address()] + list(args), 'P' + signature
        return list(args), signature

    def _init_compute(self):
        self._desc = self._module_desc()
        self.module = self.backend.build(self._desc)
        spec = self._spec
        if hasattr(self.backend, 'set_x_ghost_unused') and not self._local_periodic()[0] and \
                not getattr(self.config, 'debug_dump_dists', False) and os.environ.get('SLF_X_GHOST_STORES', '0') != '1':
            # ghost columns behind an x face that is neither periodic nor connected.  Two-copy pattern: nothing ever
            # reads them.  In-place pattern: the odd step of the first / last real column PULLS out of them what its even
            # step pushed there two steps earlier (the reference's behaviour for a wet node next to an open face), so the
            # stores may only go where that column holds no wet node -- a wall, as in every cavity / channel / pipe: what
            # a dry node pulls out of the ghost column only ever travels back into it.  (Raw dumps of the arrays keep the
            # reference's pushes: --debug_dump_dists.)
            wet = self._subdomain.fluid_map(wet=True)
            ab = self.config.access_pattern == 'AB'
            low = not spec.has_face_conn(spec.X_LOW) and (ab or not wet[..., 0].any())
            high = not spec.has_face_conn(spec.X_HIGH) and (ab or not wet[..., -1].any())
            self.backend.set_x_ghost_unused(self.module, low, high)
        self._calc_stream = self.backend.make_stream()
        # the halo stream: ahead of the bulk sweep's queued workgroups where the backend can say so
        # (not with the peer transport: its waits are kernels that spin on this stream until the neighbours have signalled,
        # and a spinning kernel in a high-priority queue holds the sweeps back -- DESIGN.md §7)
        prio = getattr(self.backend, 'supports_stream_priority', False) and os.environ.get('SLF_HALO_PRIORITY', '1') != '0' and \
            not getattr(self._connector, 'zero_copy', False)
        self._data_stream = self.backend.make_stream(high_priority=True) if prio else self.backend.make_stream()
        self._dist_stride = hipabi.dist_stride(self._desc)

    def _init_gpu_data(self):
        self._alloc_distributions()     # first: placed arrays want an allocator that has handed out nothing yet
        b = self.backend
        # fields and node map: x = 1 of every row on a 128-byte line, like the distributions (the sweeps read / store them
        # with the lane offsets of the populations: aligned accesses, 8-byte ones in the two-nodes-per-thread kernels)
        def aligned(host):
            kw = {'align_offset': b.dist_align_offset(host.dtype.itemsize)} if hasattr(b, 'dist_align_offset') else {}
            return b.alloc_buf(like=host, **kw)
        for field in self._scalar_fields:
            self._gpu_field_map[id(field)] = aligned(self._host_base[id(field)])
        for vec in self._vector_fields:
            for comp in vec:
                self._gpu_field_map[id(comp)] = aligned(self._host_base[id(comp)])
        self._gpu_geo_map = aligned(self._host_base[id(self._subdomain._type_map_ghost)])
        self.row_classes = None
        if getattr(b, 'supports_row_classes', None) and b.supports_row_classes(self._desc) and \
                getattr(self.config, 'hip_row_classes', True) and os.environ.get('SLF_ROW_CLASSES', '1') != '0':
            # which waves need the map at all, which rows need boundary-condition code (backend_hip.classify_rows)
            self.row_classes = b.classify_rows(self.module, self._gpu_geo_map, self._calc_stream)
            self.config.logger.debug('row classes: %s' % self.row_classes)
        if self.indirect:
            self._build_indirect_address_map()

    def _alloc_distributions(self):
        b = self.backend
        nbytes = self._sim.grid.Q * self._dist_stride * self.float().itemsize
        off = b.dist_align_offset(self.float().itemsize)
        ab = self.config.access_pattern == 'AB'
        from sailfish_amd import placement
        if placement.enabled() and nbytes >= placement.MIN_BYTES and not self.indirect and \
                getattr(self.config, 'hip_placement', True):
            # large arrays: physical backing spread over HBM (placement.py); all lattices and copies share the span
            n = len(self._sim.grids)
            sizes = [nbytes] * (n * (2 if ab else 1))
            cfg = self.config
            if getattr(cfg, 'hip_placement_tune', True) and os.environ.get('SLF_PLACEMENT_TUNE', '1') != '0' and \
                    not (0 < cfg.max_iters < 200) and not placement.holding_now():
                # placement by measurement: a second set is placed while the first is allocated, the faster one stays
                bufs, self.placement_tuning = placement.choose(
                    lambda: b.alloc_placed(sizes, off), lambda bs: self._probe_placement(bs, n, ab),
                    lambda bs: [b.free_buf(pb.addr) for pb in bs], log=cfg.logger.debug,
                    room=lambda: placement.room_for(b, sum(sizes)))
            else:
                bufs = b.alloc_placed(sizes, off)
            self._gpu_grids_primary = [pb.addr for pb in bufs[:n]]
            self._gpu_grids_secondary = [pb.addr for pb in bufs[n:]]
            self.config.logger.debug('placed distributions: %s' % b.last_placement)
        else:
            for _ in self._sim.grids:
                self._gpu_grids_primary.append(b.alloc_buf(size=nbytes, align_offset=off))
                if ab:
                    self._gpu_grids_secondary.append(b.alloc_buf(size=nbytes, align_offset=off))
        self.config.logger.debug('distributions: %d MiB' % (nbytes * (2 if self._gpu_grids_secondary else 1) >> 20))

    placement_tuning = None

    def _probe_placement(self, bufs, n_grids, ab, steps=12):
        """Seconds per step of the plain fluid sweep on the first lattice of the placed set `bufs`: what
        placement.choose() compares (placement.probe_sweep)."""
        from sailfish_amd import placement
        nbytes = self._sim.grid.Q * self._dist_stride * self.float().itemsize
        src, dst = bufs[0].addr, (bufs[n_grids].addr if ab else bufs[0].addr)
        return placement.probe_sweep(self.backend, self._desc, self.dim, src, dst, nbytes, self._calc_stream, steps)

    def gpu_field(self, field):
        if isinstance(field, list):
            return [self._gpu_field_map[id(f)] for f in field]
        return self._gpu_field_map[id(field)]

    def gpu_dist(self, num, copy):
        """Device address of lattice `num`, copy 0 (A) or 1 (B; the same buffer as A in the AA pattern)."""
        if copy == 0:
            return self._gpu_grids_primary[num]
        if self._gpu_grids_secondary:
            return self._gpu_grids_secondary[num]
        return self._gpu_grids_primary[num]

    def gpu_geo_map(self):
        return self._gpu_geo_map

    @property
    def gpu_scratch_space(self):
        return None

    def get_kernel(self, name, args, args_format, block_size=None, needs_iteration=False, shared=0,
                   more_shared=False):
        return self.backend.get_kernel(self.module, name, block=block_size or (self.config.block_size,),
                                       args=[int(a) for a in args], args_format=args_format, shared=shared,
                                       needs_iteration=needs_iteration)

    def exec_kernel(self, name, args, args_format, needs_iteration=False):
        kernel = self.get_kernel(name, args, args_format, needs_iteration=needs_iteration)
        self.backend.run_kernel(kernel, None, self._calc_stream)

    # ------------------------------------------------------------------ halo
    def _init_halo(self):
        """Index lists, device buffers and pack / unpack kernels for every neighbour.  A model with
        several lattices (binary fluids) sends them back to back in one message per neighbour."""
        self._links = {}
        self._macro_links = {}
        self._ev_halo = None
        if self._all_specs is None or len(self._all_specs) < 2:
            return
        if self._connector is None:
            from sailfish_amd.connector import LocalConnector
            self._connector = LocalConnector()
        if self._x_faces_only():
            return self._init_xface_halo()
        arr = list(reversed(self._physical_size))
        dense_nodes = self._get_nodes()
        links = subdomain_connection.build_halo_links(self._spec, self._all_specs, self._global_size,
                                                      self._global_periodic, self._sim.grid, arr,
                                                      dense_nodes if self.indirect else self._dist_stride,
                                                      fused=self._fused)
        if self.indirect:
            self._translate_halo_links(links, self._host_indirect_address, dense_nodes, self._dist_stride)
        b = self.backend
        if self._connector is None:
            from sailfish_amd.connector import LocalConnector
            self._connector = LocalConnector()
        aa = self.config.access_pattern == 'AA'
        modes = ('push', 'pull') if aa else ('push',)
        n_grids = len(self._gpu_grids_primary)
        isz = np.dtype(self.float).itemsize
        # zero-copy connectors (connector.PeerConnector): my pack kernels write the neighbour's receive buffer, which is
        # mapped here; two sets that alternate by step parity
        zc = getattr(self._connector, 'zero_copy', False)
        todo = []
        for nid in sorted(links):
            link = links[nid]
            n_send = max(len(link.push_send), len(link.pull_send))
            n_recv = max(len(link.push_recv), len(link.pull_recv))
            if n_send == 0 and n_recv == 0:
                continue
            if zc:
                link.recv_bufs = [self._connector.alloc_recv(self, 'dist', nid, par, n_recv * n_grids, self.float) for par in (0, 1)]
            else:
                link.send_buf = self._connector.alloc_buffer(self, n_send * n_grids, self.float)
                link.recv_buf = self._connector.alloc_buffer(self, n_recv * n_grids, self.float)
                link.send_bufs, link.recv_bufs = [link.send_buf] * 2, [link.recv_buf] * 2
            todo.append((nid, link))
        if zc:
            self._connector.resolve(self)            # collective: every rank, whatever its links
        for nid, link in todo:
            if zc:
                link.send_bufs = [self._connector.send_addr(self, 'dist', nid, par) for par in (0, 1)]
                link.send_buf, link.recv_buf = link.send_bufs[0], link.recv_bufs[0]
            link.kernels = {}
            for mode in modes:
                s_idx = getattr(link, mode + '_send')
                r_idx = getattr(link, mode + '_recv')
                g_s = b.alloc_buf(like=s_idx) if len(s_idx) else 0
                g_r = b.alloc_buf(like=r_idx) if len(r_idx) else 0
                for copy in range(1 if not self._gpu_grids_secondary else 2):
                    # the parity of the steps these kernels serve: in place the even steps pull, two-copy the steps that
                    # write copy `copy`
                    par = (0 if mode == 'pull' else 1) if aa else 1 - copy
                    packs, unpacks = [], []
                    for g in range(n_grids):
                        dist = self.gpu_dist(g, copy)
                        if len(s_idx):
                            packs.append(self.get_kernel('CollectSparseData', [
                                g_s, dist, link.send_bufs[par] + g * len(s_idx) * isz, len(s_idx)], 'PPPi'))
                        if len(r_idx):
                            unpacks.append(self.get_kernel('DistributeSparseData', [
                                g_r, dist, link.recv_bufs[par] + g * len(r_idx) * isz, len(r_idx)], 'PPPi'))
                    link.kernels[(mode, copy)] = (packs, unpacks, len(s_idx) * n_grids, len(r_idx) * n_grids)
            self._links[nid] = link

    # -- 1-D decompositions along x: dense face buffers written / read by the sweep itself (xface.py)
    _xface = None

    def _x_faces_only(self):
        """Every subdomain of the simulation is connected through its x faces only, and the model can use the
        x-face buffers (same answer in every runner of the simulation)."""
        if not getattr(self.config, 'hip_xface', True) or self.has_macro_exchange or self.dim != 3 or \
                not getattr(self.backend, 'supports_xface', False):
            return False
        if not xface.supported(self._sim.grid, self._desc, self.indirect) or len(self._sim.grids) != 1:
            return False
        return self._x_slabs_line_up()

    def _x_slabs_line_up(self):
        """A 1-D decomposition along x whose slabs share their y / z extent (same answer in every runner)."""
        local = self._local_periodic()
        if any(local[a] and not self._fused[a] for a in range(self.dim)):
            # periodic images made by the ghost-layer kernels live in the arrays, not in the face buffers
            return False
        ref = self._all_specs[0]
        for spec in self._all_specs:
            if tuple(spec.location[1:]) != tuple(ref.location[1:]) or tuple(spec.size[1:]) != tuple(ref.size[1:]):
                return False        # faces that only partly overlap: rows of the two sides do not line up
            faces = set(face for face, _ in spec.connecting_subdomains())
            if not faces or not faces <= set((spec.X_LOW, spec.X_HIGH)):
                return False
            if spec.size[0] > 1024 or spec.size[0] < 2:
                return False
        return True

    def _init_xface_halo(self):
        """Per neighbour and step parity one send and one receive buffer: [what leaves through my low face | through my
        high face] (the faces that lead to this neighbour); it arrives as [its high-face input | its low-face input] on
        the other side.  The sweep is cut into z-chunks and the planes a chunk completes travel at once
        (xface.ChunkPlan)."""
        spec = self._spec
        n = xface.face_count(self._desc)
        isz = np.dtype(self.float).itemsize
        by_neighbour = {}
        for face, nid in sorted(spec.connecting_subdomains()):
            by_neighbour.setdefault(nid, []).append(xface.LOW if face == spec.X_LOW else xface.HIGH)
        send, recv = [[0, 0], [0, 0]], [[0, 0], [0, 0]]
        self._xface_routes = []          # (neighbour id, my face, parity -> send address, parity -> receive address)
        zc = getattr(self._connector, 'zero_copy', False)
        for nid in sorted(by_neighbour):
            faces = sorted(by_neighbour[nid])
            link = subdomain_connection.HaloLink(nid)
            link.faces = faces
            if zc:
                link.recv_bufs = [self._connector.alloc_recv(self, 'dist', nid, par, n * len(faces), self.float) for par in (0, 1)]
            else:
                link.send_bufs = [self._connector.alloc_buffer(self, n * len(faces), self.float) for _ in (0, 1)]
                link.recv_bufs = [self._connector.alloc_buffer(self, n * len(faces), self.float) for _ in (0, 1)]
            self._links[nid] = link
        if zc:
            # my send planes ARE the neighbours' receive planes: [through my low face | through my high face] on this side
            # is [its high-face input | its low-face input] on the other, the layout of its receive buffer
            self._connector.resolve(self)
            for nid, link in self._links.items():
                link.send_bufs = [self._connector.send_addr(self, 'dist', nid, par) for par in (0, 1)]
        for nid in sorted(by_neighbour):
            link, faces = self._links[nid], self._links[nid].faces
            link.send_buf, link.recv_buf = link.send_bufs[0], link.recv_bufs[0]
            for par in (0, 1):
                for k, face in enumerate(faces):                       # my send order: low, high
                    send[par][face] = link.send_bufs[par] + k * n * isz
                for k, face in enumerate(reversed(faces)):             # the neighbour's send order seen from here
                    recv[par][face] = link.recv_bufs[par] + k * n * isz
            for face in faces:            # pieces are posted in this order on both sides: my low <-> its high first
                self._xface_routes.append((nid, face))
            link.kernels = {}
            for mode in ('push', 'pull'):
                for copy in (0, 1):
                    link.kernels[(mode, copy)] = ([], [], n * len(faces), n * len(faces))
        self._xface = xface.XFaceHalo(self.backend, self.module, self._sim.grid, self._desc, send, recv, shared=zc)
        lat = list(reversed(self._lat_size))
        # several launches per step pay off where the transfer is slow (another process / GPU: a connector that can be
        # called in the middle of a step); runners stepped in lock-step by one Python process are bound by that process
        # instead: one chunk (profiles/r03/xface_overlap_schemes.jsonl)
        self._xchunks = xface.ChunkPlan(lat[2] - 2, self._fused[2],
                                        None if getattr(self._connector, 'mid_step', False) else 1)

    def xface_pieces(self, pos):
        """[(neighbour id, send address, receive address, elements)] of batch `pos` of the step just enqueued: for every
        connected face (send side: my faces low, high; the receive side of the same neighbour takes them as its high, low)
        the runs of z-planes the chunks swept so far have completed."""
        x, isz = self._xface, np.dtype(self.float).itemsize
        par = self._xface_parity
        out = []
        # sends in my face order; receives of one neighbour in the order IT sends: its low face (= my high) first
        by_n = {}
        for nid, face in self._xface_routes:
            by_n.setdefault(nid, []).append(face)
        for nid in sorted(by_n):
            s_faces = sorted(by_n[nid])
            r_faces = list(reversed(s_faces))
            for sf, rf in zip(s_faces, r_faces):
                for p0, p1 in self._xchunks.batches[self._xface_kind][pos]:
                    off, cnt = p0 * x.plane * isz, (p1 - p0) * x.plane
                    out.append((nid, x.send[par][sf] + off, x.recv[par][rf] + off, cnt))
        return out

    def _reset_xface(self):
        """Whatever state was just set up from the host (initial conditions, a checkpoint, a debug write): the arrays
        count, nothing that crossed the x faces before does."""
        if self._xface is not None:
            self.backend.sync_stream(*self._all_streams())
            self._connector.quiesce(self)       # peer transport: the neighbours write into these buffers themselves
            self._xface.reset(self._calc_stream)
            it = self._sim.iteration
            if self.config.access_pattern == 'AA' and (it & 1):
                # the next step pulls, and the edge lanes of the fluid-only row kernel take what enters through a
                # connected x face from the receive buffers alone (slf_row.hip: no pull out of the ghost column): prime
                # them from the ghost columns of the state just written (a checkpoint taken at an odd iteration)
                self._xface.prime_pull(self.gpu_dist(0, 0), self._calc_stream, parity=1 - (it & 1))
            self.backend.sync_stream(self._calc_stream)
            self._connector.quiesce(self)
            self.__dict__.pop('_halo_mode', None)
    
    def _materialise_halo(self):
        """x-face buffers: the arrays are stale at the connected faces until the received values are written into
        them (before anything reads the arrays on the host)."""
        if self._xface is None or not hasattr(self, '_halo_mode'):
            return
        self.backend.sync_stream(*self._all_streams())
        self._xface.materialise(self.gpu_dist(0, self._halo_copy), self._halo_mode == 'push', self._calc_stream,
                                parity=self._xface_parity)
        self.backend.sync_stream(self._calc_stream)

    def halo_messages(self, kind='dist'):
        """[(neighbour id, send buffer, #send, receive buffer, #recv)] of the exchange that is due now,
        ordered by neighbour id: 'dist' = populations of the step just computed, 'macro' = fields read
        by non-local models."""
        out = []
        if kind == 'dist':
            for nid in sorted(self._links):
                link = self._links[nid]
                k = link.kernels[(self._halo_mode, self._halo_copy)]
                out.append((nid, link.send_buf, k[2], link.recv_buf, k[3]))
        else:
            for nid in sorted(self._macro_links):
                link = self._macro_links[nid]
                out.append((nid, link.send_buf, link.n_send, link.recv_buf, link.n_recv))
        return out

    # ------------------------------------------------------------------ kernels
    def _prepare_compute_kernels(self):
        self._kernels_full = self._sim.get_compute_kernels(self, True, True)
        self._kernels_none = self._sim.get_compute_kernels(self, False, True)
        self._pbc_kernels = self._sim.get_pbc_kernels(self)
        self._pbc_axes = [a for a in range(self.dim) if self._local_periodic()[a] and not self._fused[a]]
        self._regions = self._make_regions()
        self._kernels_prepared = True

    def _make_regions(self):
        """(boundary regions, bulk region): rows / planes next to faces that exchange halos are swept
        first so that packing and the transfer overlap the bulk sweep (reference subdomain_runner.py:
        1028-1058).  x faces cannot be split off (a workgroup owns whole rows)."""
        lat = list(reversed(self._lat_size))
        ny = lat[1] - 2
        nz = lat[2] - 2 if self.dim == 3 else 1
        y0, y1 = 1, ny + 1
        z0, z1 = (1, nz + 1) if self.dim == 3 else (0, 1)
        full = (y0, y1, z0, z1)
        spec = self._spec
        if not self._links or not getattr(self.config, 'bulk_boundary_split', True):
            return [], full
        if spec.has_face_conn(spec.X_LOW) or spec.has_face_conn(spec.X_HIGH):
            return [], full
        z_conn = self.dim == 3 and (spec.has_face_conn(spec.Z_LOW) or spec.has_face_conn(spec.Z_HIGH))
        y_conn = spec.has_face_conn(spec.Y_LOW) or spec.has_face_conn(spec.Y_HIGH)
        if (z_conn and nz <= 2) or (y_conn and ny <= 2):
            # a connected face whose layer cannot be split off: its ghosts would be written by the bulk launch,
            # after the event the pack waits for -- sweep everything first
            return [], full
        bnd = []
        bz0, bz1 = z0, z1
        if self.dim == 3:
            if spec.has_face_conn(spec.Z_LOW):
                bnd.append((y0, y1, 1, 2))
                bz0 = 2
            if spec.has_face_conn(spec.Z_HIGH):
                bnd.append((y0, y1, nz, nz + 1))
                bz1 = nz
        by0, by1 = y0, y1
        if spec.has_face_conn(spec.Y_LOW):
            bnd.append((1, 2, bz0, bz1))
            by0 = 2
        if spec.has_face_conn(spec.Y_HIGH):
            bnd.append((ny, ny + 1, bz0, bz1))
            by1 = ny
        if not bnd:
            return [], full
        return bnd, (by0, by1, bz0, bz1)

    # ------------------------------------------------------------------ stepping
    has_macro_exchange = False

    def step(self, sync_req=False):
        """One time step of a runner that owns its process (neighbours, if any, live in other processes): the step is
        ONE program (_program, stepqueue.py), replayed from a C-ABI step plan -- a single runtime call -- where the
        transport can be part of it (RCCL through the C ABI), performed entry by entry otherwise (torch.distributed) and
        on the steps that record timing events."""
        b = self.backend
        it = self._sim.iteration
        self._update_dynamic_params(it)
        if self._plan_ok and not self._profile.wants_gpu_events():
            key = (it & 1, bool(sync_req))
            plan = self._plans.get(key)
            if plan is None:
                plan = b.make_plan()
                try:
                    self._program(plan, it, sync_req)
                    self._plans[key] = plan
                except NotPlannable:            # the transport needs Python between the launches
                    self._plan_ok, plan = False, None
            if plan is not None:
                self._set_step_state(it)
                plan.run(it)
                if self._xface is not None:
                    self._xface._bound = None       # the plan set the module's face buffers itself
                self._sim.iteration += 1
                b.set_iteration(self._sim.iteration)
                return
        b.set_iteration(it)
        self._program(DirectQueue(b), it, sync_req)
        if self._xface is not None:
            self._xface._bound = None
        self._sim.iteration += 1
        b.set_iteration(self._sim.iteration)

    def _time_dependent(self):
        enc = getattr(self._subdomain, '_encoder', None)
        return (enc is not None and getattr(enc, 'time_dependent', False)) or getattr(self._sim, 'time_dependent_force', False)

    def _update_dynamic_params(self, it):
        """Boundary values that depend on time (node_type.DynamicValue with sym.S.time / time series; reference: device
        code, boundary.mako:52-84): evaluated on the host for LB iteration `it` and written into the kernels' parameter
        table on the calc stream, in front of the step's launches (C ABI slf_module_update_node_params; a one-wave kernel
        whose arguments ARE the values: the host does not wait)."""
        if self._time_dependent():
            for first, values in self._subdomain._encoder.dynamic_updates(it):
                self.backend.update_node_params(self.module, first, values, self._calc_stream)
            if getattr(self._sim, 'time_dependent_force', False):
                # the acceleration is an argument of the sweep launches: the ones enqueued from here on take this value
                self.backend.set_body_force(self.module, self._sim.body_force_at(it))

    # ------------------------------------------------------------------ the step as a program (stepqueue.py)
    _plan_ok = False

    def _init_step_program(self):
        """Events, streams and plan table of step() (runners that own their process; a same-process group keeps the
        three-phase step_compute / exchange / step_finish protocol of controller.LocalGroup)."""
        b = self.backend
        self._plans = {}
        self._plan_ok = bool(getattr(b, 'supports_step_plans', False)) and getattr(self.config, 'hip_step_plans', True) and \
            os.environ.get('SLF_STEP_PLAN', '1') != '0'
        names = ('bnd', 'bulk', 'halo', 'packed', 'copied', 'macro', 'macro_halo', 'macro_packed', 'macro_copied')
        self._pev = [dict((n, b.make_event(self._calc_stream)) for n in names) for _ in (0, 1)]
        # a second calc stream (SLF_CALC_STREAMS=1: off).  z / y decompositions sweep their face layers on it: the
        # interior depends on the neighbours only through the face layers of the step before, so its stream never waits
        # for a transfer.  x decompositions alternate their z-chunks between the two streams: a chunk starts while the
        # one before it drains.  Ghost-layer PBC kernels and the macro pass of the non-local models work on whole
        # arrays: one stream there.
        two = os.environ.get('SLF_CALC_STREAMS', '2') != '1' and not self._pbc_axes and not self.has_macro_exchange and \
            bool(self._links) and hasattr(b, 'supports_step_plans') and not self._time_dependent()
        # (time-dependent boundary values are rewritten on the calc stream before every step: every sweep launch on it)
        if self._xface is not None:
            # z-chunks on alternating streams only on request: measured slower than the drain at the chunk boundaries
            # it avoids (profiles/r04/NOTES.md)
            two = two and not self._xface.needs_clear and os.environ.get('SLF_XFACE_STREAMS', '1') == '2'
            n = len(self._xchunks.order)
            self._ev_chunk = [[b.make_event(self._calc_stream) for _ in range(n)] for _ in (0, 1)]
            self._ev_batch = [[b.make_event(self._data_stream) for _ in range(n)] for _ in (0, 1)]
        else:
            two = two and bool(self._regions[0])
        self._bnd_stream = b.make_stream() if two else self._calc_stream

    def _sweep_kernels(self, it, sync_req):
        kernels = self._kernels_full if sync_req else self._kernels_none
        return kernels.primary if (it & 1) == 0 else kernels.secondary

    def _set_step_state(self, it):
        """What the rest of the runner reads about the step being enqueued (halo_messages, xface_pieces, materialise)."""
        aa = self.config.access_pattern == 'AA'
        self._halo_mode = 'pull' if (aa and (it & 1) == 0) else 'push'
        self._halo_copy = 0 if aa else 1 - (it & 1)
        if self._xface is not None:
            self._xface_parity = it & 1
            self._xface_kind = 'own' if (aa and (it & 1) == 0) else 'push'

    def _program(self, q, it, sync_req):
        """One step of a runner that owns its process (reference subdomain_runner.py:960-1058): macro pass of the
        non-local models, sweep, halo -- the exchanges through the connector."""
        self._set_step_state(it)
        self._program_macro(q, it, sync_req=sync_req)
        self._program_front(q, it, sync_req)
        self._program_back(q, it)

    def _neighbour_events(self, group, name, parity):
        """Events `name` of parity `parity` of the runners this one exchanges halos with (same-process groups)."""
        ids = set(self._links) | set(getattr(self, '_macro_links', {}))
        return [group.by_id[nid]._pev[parity][name] for nid in sorted(ids) if nid in group.by_id]

    def _program_macro(self, q, it, group=None, sync_req=False):
        """Macroscopic-field pass of the non-local models (NNSubdomainRunner); nothing here."""

    def _program_macro_back(self, q, it):
        pass

    def _program_front(self, q, it, sync_req, group=None):
        """[face layers -> event] -> interior -> ghost-layer PBC kernels on the calc stream(s); pack on the data stream.
        group: the controller.LocalGroup that steps this runner together with its neighbours in one process -- the group
        moves the packed buffers itself, between the fronts and the backs of its runners; None: the connector does, right
        here (x-face pieces: after every z-chunk)."""
        prof, timed = self._profile, not q.planned
        ev, pev = self._pev[it & 1], self._pev[1 - (it & 1)]
        kernels = self._sweep_kernels(it, sync_req)
        if self._xface is not None:
            return self._program_xface(q, it, kernels, group)
        sk, sb = self._calc_stream, self._bnd_stream
        bnd, bulk = self._regions
        ready = None
        if bnd:
            if self._links:
                q.wait(sb, pev['halo'])
            if sb is not sk:
                q.wait(sb, pev['bulk'])
            if timed:
                prof.record_gpu_start(TimeProfile.BOUNDARY, sb)
            for reg in bnd:
                for k in kernels:
                    q.launch(k, reg, sb)
            if timed:
                prof.record_gpu_end(TimeProfile.BOUNDARY, sb)
            q.record(ev['bnd'], sb)
            if sb is not sk:
                q.wait(sk, pev['bnd'])
            ready = ev['bnd']
        elif self._links:
            q.wait(sk, pev['halo'])
        self._program_sweep_rest(q, it, kernels, bulk, ready, group)
        if bnd and sb is not sk and sync_req:
            # the host is about to look at this step's results (fields to the host, the invalid-value poll, kernels a
            # simulation enqueues from after_step() -- all on the calc stream): they come after the face layers too
            q.wait(sk, ev['bnd'])

    def _program_sweep_rest(self, q, it, kernels, bulk, ready, group):
        prof, timed = self._profile, not q.planned
        ev = self._pev[it & 1]
        sk = self._calc_stream
        if timed:
            prof.record_gpu_start(TimeProfile.BULK, sk)
        for k in kernels:
            q.launch(k, bulk, sk)
        if timed:
            prof.record_gpu_end(TimeProfile.BULK, sk)
        base = 1 - (it & 1)
        for axis in self._pbc_axes:
            for k in self._dist_pbc_kernels()[base][axis]:
                q.launch(k, None, sk)
        q.record(ev['bulk'], sk)
        if ready is None or self._pbc_axes:
            ready = ev['bulk']       # unsplit subdomains: the whole sweep (and the local PBC) must be done first
        self._program_pack(q, it, ready, group)

    def _dist_pbc_kernels(self):
        return self._pbc_kernels

    def _program_pack(self, q, it, ready, group, kind='dist'):
        """Data stream: wait(`ready`) -> pack -> event 'packed'; without a group the exchange follows at once."""
        links = self._links if kind == 'dist' else self._macro_links
        if not links:
            return
        prof, timed, sh = self._profile, not q.planned, self._data_stream
        ev = self._pev[it & 1]
        pre = '' if kind == 'dist' else 'macro_'
        q.wait(sh, ready)
        if group is not None:
            # a send buffer is not written again before the neighbours' copies of the previous step have read it
            for e in self._neighbour_events(group, pre + 'copied', 1 - (it & 1)):
                q.wait(sh, e)
        tp = TimeProfile.COLLECTION if kind == 'dist' else TimeProfile.MACRO_COLLECTION
        if timed:
            prof.record_gpu_start(tp, sh)
        for nid in sorted(links):
            link = links[nid]
            packs = link.kernels[(self._halo_mode, self._halo_copy)][0] if kind == 'dist' else link.packs[it & 1]
            for pack in packs:
                q.launch(pack, None, sh)
        if timed:
            prof.record_gpu_end(tp, sh)
        q.record(ev[pre + 'packed'], sh)
        if group is None:
            if timed and kind == 'dist':
                prof.record_cpu_start(TimeProfile.RECV_DISTS)
            self._connector.enqueue_exchange(q, self, kind)
            if timed and kind == 'dist':
                prof.record_cpu_end(TimeProfile.RECV_DISTS)

    def _program_back(self, q, it):
        """Data stream: unpack what the exchange delivered -> event 'halo' (the next step's face layers wait for it)."""
        if not self._links or self._xface is not None:
            return
        prof, timed, sh = self._profile, not q.planned, self._data_stream
        if timed:
            prof.record_gpu_start(TimeProfile.DISTRIB, sh)
        for nid in sorted(self._links):
            for unpack in self._links[nid].kernels[(self._halo_mode, self._halo_copy)][1]:
                q.launch(unpack, None, sh)
        if timed:
            prof.record_gpu_end(TimeProfile.DISTRIB, sh)
        q.record(self._pev[it & 1]['halo'], sh)

    def _program_xface(self, q, it, kernels, group=None):
        """1-D x decomposition: the sweep in z-chunks; after each chunk the planes of the face buffers that are now
        complete travel on the data stream (own process: through the connector, right here; same-process group: the
        group copies them after the fronts of all its runners).  A chunk waits for the transfers of the previous step
        that carry the planes it reads (xface.ChunkPlan); with SLF_XFACE_STREAMS=2 the chunks alternate between the two
        calc streams and also wait for the chunks of the previous step that touched their planes or their neighbours."""
        prof, timed = self._profile, not q.planned
        x, plan = self._xface, self._xchunks
        par = it & 1
        kind = self._xface_kind
        prev_kind = 'push' if (self.config.access_pattern != 'AA' or kind == 'own') else 'own'
        ny = list(reversed(self._lat_size))[1] - 2
        streams = [self._calc_stream, self._bnd_stream]
        snd, rcv = x.send[par], x.recv[1 - par]
        # every sweep of the group on one stream (controller.LocalGroup._serialise_sweeps): its order is all the order
        # the shared face buffers need -- no events
        serial = group is not None and getattr(group, 'single_calc_stream', False) and x.shared
        if group is not None and not serial:
            # the send set of this parity was last read by the neighbours' copies of step it - 2
            for e in self._neighbour_events(group, 'copied', par):
                q.wait(streams[0], e)
        q.xface(self.module, snd[xface.LOW], snd[xface.HIGH], rcv[xface.LOW], rcv[xface.HIGH])
        if x.needs_clear:
            for a in snd:
                if a:
                    q.memset(a, 0xFF, x.nbytes, streams[0])
        evc, evb = self._ev_chunk[par], self._ev_batch[par]
        pevc, pevb = self._ev_chunk[1 - par], self._ev_batch[1 - par]
        # face buffers that are not copied (the neighbour's receive planes ARE my send planes: peer transport, subdomains of
        # one process): a chunk also waits until the planes it writes have been read (xface.ChunkPlan.peer_need)
        need = plan.peer_need(kind, prev_kind) if x.shared else plan.need[prev_kind]
        every = x.shared and group is None      # a signal after every chunk: the neighbours' chunks count on it
        pos_of = dict((c, pos) for pos, c in enumerate(plan.order))
        sh = self._data_stream
        if timed:
            prof.record_gpu_start(TimeProfile.BULK, streams[0])
        waited = {}
        for pos, c in enumerate(plan.order):
            st = streams[pos & 1]
            if serial:
                for k in kernels:
                    q.launch(k, plan.region(c, ny), st)
                continue
            if need[c] > waited.get(id(st), -1):     # the streams are in order: a later transfer waited for covers the earlier ones
                q.wait(st, pevb[need[c]])
                waited[id(st)] = need[c]
            for c2 in plan.neighbours(c):
                if streams[pos_of[c2] & 1] is not st:
                    q.wait(st, pevc[pos_of[c2]])
            for k in kernels:
                q.launch(k, plan.region(c, ny), st)
            if not plan.exchanges_at(pos) and streams[0] is streams[1] and not every:
                continue                 # nothing travels after this chunk and nobody waits for it
            q.record(evc[pos], st)
            if group is None:
                q.wait(sh, evc[pos])
                self._connector.enqueue_pieces(q, self, self.xface_pieces(pos))
                q.record(evb[pos], sh)
        if timed:
            if streams[1] is not streams[0]:
                streams[0].wait_for_event(evc[len(plan.order) - 1])
            prof.record_gpu_end(TimeProfile.BULK, streams[0])

    # ------------------------------------------------------------------ data movement
    def _fields_to_host(self, sync=True):
        bs = getattr(self, '_bnd_stream', None)
        if bs is not None and bs is not self._calc_stream:
            bs.synchronize()            # the face layers / every other z-chunk stored their fields on the second stream
        for field in self._scalar_fields:
            self.backend.from_buf_async(self.gpu_field(field), self._calc_stream)
        for vec in self._vector_fields:
            for comp in vec:
                self.backend.from_buf_async(self.gpu_field(comp), self._calc_stream)
        if sync:
            self.backend.sync_stream(self._calc_stream)

    def _fields_to_gpu(self):
        for field in self._scalar_fields:
            self.backend.to_buf(self.gpu_field(field))
        for vec in self._vector_fields:
            for comp in vec:
                self.backend.to_buf(self.gpu_field(comp))

    def _debug_get_dist(self, output=True, grid_num=0, copy=None):
        """Distributions as [Q, (nz,) ny, arr_nx] (reference subdomain_runner.py:1363-1381); with indirect
        addressing the slots are scattered back to their nodes (inactive nodes: 0)."""
        self.backend.sync_stream(*self._all_streams())
        self._materialise_halo()
        if copy is None:
            copy = 0 if not self._gpu_grids_secondary else (self._sim.iteration & 1)
        raw = np.zeros((self._sim.grid.Q, self._dist_stride), dtype=self.float)
        self.backend.from_buf(self.gpu_dist(grid_num, copy), raw)
        if self.indirect:
            addr = self._host_indirect_address.reshape(-1)
            act = addr != hipabi.SLF_INVALID_NODE
            dense = np.zeros((self._sim.grid.Q, self._get_nodes()), dtype=self.float)
            dense[:, act] = raw[:, addr[act]]
            return dense.reshape([self._sim.grid.Q] + self._physical_size)
        return np.ascontiguousarray(raw[:, :self._get_nodes()]).reshape([self._sim.grid.Q] + self._physical_size)

    def _debug_set_dist(self, dbuf, output=True, grid_num=0, copy=None):
        if copy is None:
            copy = 0 if not self._gpu_grids_secondary else (self._sim.iteration & 1)
        raw = np.zeros((self._sim.grid.Q, self._dist_stride), dtype=self.float)
        dense = np.asarray(dbuf, dtype=self.float).reshape(self._sim.grid.Q, -1)
        if self.indirect:
            addr = self._host_indirect_address.reshape(-1)
            act = addr != hipabi.SLF_INVALID_NODE
            raw[:, addr[act]] = dense[:, act]
        else:
            raw[:, :self._get_nodes()] = dense
        self.backend.to_buf(self.gpu_dist(grid_num, copy), raw)
        if isinstance(getattr(self, '_resident', None), dict):
            self._resident['equalised'] = False      # the scratch copies of the resident launches no longer agree with the arrays
        self._reset_xface()

    def _debug_global_idx_to_tuple(self, gi):
        dist_num = gi // self._get_nodes()
        dist_idx = gi % self._get_nodes()
        return (dist_num,) + tuple(np.unravel_index(dist_idx, self._physical_size))

    def save_checkpoint(self):
        """<base>.<iter>.<subdomain>.cpoint.npz with the simulation state and dist<N>a[/dist<N>b] of every
        lattice N (reference :1414-1431; --single_checkpoint keeps one file that is overwritten)."""
        if getattr(self.config, 'single_checkpoint', False):
            fname = io.checkpoint_filename(self.config.checkpoint_file, 1, self._spec.id, 0)
        else:
            fname = io.checkpoint_filename(self.config.checkpoint_file, io.filename_iter_digits(self.config.max_iters),
                                           self._spec.id, self._sim.iteration)
        data = {'state': np.frombuffer(pickle.dumps(self._sim.get_state()), dtype=np.uint8)}
        cur = (self._sim.iteration & 1) if self._gpu_grids_secondary else 0
        for n in range(len(self._gpu_grids_primary)):
            # dist<N>a = gpu_dist(N, iteration & 1): the current state; dist<N>b the other copy (reference :1420-1427)
            data['dist%da' % n] = self._debug_get_dist(grid_num=n, copy=cur)
            if self._gpu_grids_secondary:
                data['dist%db' % n] = self._debug_get_dist(grid_num=n, copy=1 - cur)
        np.savez(fname, **data)

    def restore_checkpoint(self, fname):
        self.config.logger.info('Restoring checkpoint from {0}'.format(fname))
        cpoint = np.load(fname, allow_pickle=False)
        state = pickle.loads(cpoint['state'].tobytes())
        saved_it = int(state.get('iteration', 0)) if isinstance(state, dict) else 0
        # the simulation state is always restored (subclass state lives there); without --restore_time only the
        # clock is reset afterwards (reference :1436-1449)
        self._sim.set_state(state)
        if not getattr(self.config, 'restore_time', True):
            self._sim.iteration = 0
        it = self._sim.iteration
        if not self._gpu_grids_secondary and (saved_it & 1) != (it & 1):
            raise ValueError('AA checkpoint written at iteration %d cannot be continued from iteration %d: the in-place '
                             'pattern stores the populations in a layout that depends on the parity of the step'
                             % (saved_it, it))
        cur = (it & 1) if self._gpu_grids_secondary else 0
        for key in cpoint.files:
            if not key.startswith('dist'):
                continue
            n, is_a = int(key[4:-1]), key.endswith('a')
            if n >= len(self._gpu_grids_primary) or (not is_a and not self._gpu_grids_secondary):
                continue
            self._debug_set_dist(cpoint[key], grid_num=n, copy=cur if is_a else 1 - cur)
        self.backend.set_iteration(self._sim.iteration)

    # ------------------------------------------------------------------ life cycle
    def sighup_handler(self, signum, frame):
        self.config.logger.info('Received HUP signal, will save checkpoint (it=%d).' % self._sim.iteration)
        self._checkpoint_req = True

    def _install_signal_handlers(self):
        import signal
        import threading
        if hasattr(signal, 'SIGHUP') and threading.current_thread() is threading.main_thread():
            previous = signal.getsignal(signal.SIGHUP)

            def handler(signum, frame):
                self.sighup_handler(signum, frame)
                if callable(previous):          # several runners in one process: every one gets the request
                    previous(signum, frame)
            signal.signal(signal.SIGHUP, handler)

    def prepare(self):
        """Everything up to (not including) the main loop (reference run(), :1537-1602)."""
        cfg = self.config
        self._install_signal_handlers()
        self._init_geometry()
        self._sim.init_fields(self)
        self._init_compute()
        self._subdomain.init_fields(self._sim)
        self._sim.verify_fields()
        self._init_gpu_data()
        self._init_halo()
        self._prepare_compute_kernels()
        self._init_step_program()
        self._sim.initial_conditions(self)
        self.backend.set_iteration(0)
        if self._output is not None:
            self._output.set_fluid_map(self._subdomain.fluid_map())
        if getattr(cfg, 'restore_from', ''):
            fname = io.subdomain_checkpoint(cfg.restore_from, self._spec.id)
            self.restore_checkpoint(fname)
        if getattr(cfg, 'debug_dump_node_type_map', False) and self._output is not None:
            self._output.dump_node_type(self._subdomain._type_vis_map)
        self._reset_xface()
        self._sim.before_main_loop(self)
        self.backend.sync_stream(self._calc_stream)
        self.num_fluid_nodes = self._subdomain.num_fluid_nodes

    def need_quit(self):
        cfg = self.config
        if cfg.max_iters > 0 and self._sim.iteration >= cfg.max_iters:
            return True
        return self._quit_event is not None and self._quit_event.is_set()

    def pre_step(self):
        """Returns (sync_req, output_req) for the coming step (reference main(), :1668-1712)."""
        output_req = self._sim.need_output()
        last = self.config.max_iters > 0 and self._sim.iteration + 1 >= self.config.max_iters
        sync_req, fields_req = self._sim.need_sync_fields()
        if last:
            sync_req = fields_req = True
        return sync_req, fields_req, output_req

    def check_gpu_invalid(self):
        """On-GPU invalid value check (reference --check_invalid_results_gpu): the sweeps flag wet nodes with
        a non-finite density; polled whenever the host synchronises with the device anyway."""
        if not getattr(self.config, 'check_invalid_results_gpu', False):
            return
        pos = self.backend.poll_invalid(self.module, self._calc_stream)
        if pos is not None:
            gpos = [int(p) - 1 + o for p, o in zip(pos, self._spec.location)]
            raise self.backend.FatalError(
                'Invalid value (inf / nan) detected on the GPU: subdomain %d, node %s (global position %s), '
                'before iteration %d' % (self._spec.id, tuple(pos[:self.dim]), tuple(gpos), self._sim.iteration))

    def post_step(self, sync_req, output_req):
        cfg = self.config
        if sync_req:
            self.check_gpu_invalid()
            self._fields_to_host(True)
            if getattr(cfg, 'check_invalid_results_host', True) and self._output is not None and \
                    self._output._fluid_map is not None and not self._output.verify():
                raise RuntimeError('Invalid value detected in output for iteration %d' % self._sim.iteration)
        if output_req and cfg.output_required and self._output is not None:
            if cfg.output:
                self._output.save(self._sim.iteration)
            if getattr(cfg, 'debug_dump_dists', False):
                self._output.dump_dists([self._debug_get_dist()], self._sim.iteration)
        if (self._sim.need_checkpoint() or self._checkpoint_req) and cfg.checkpoint_file:
            self._checkpoint_req = False
            self.save_checkpoint()
        self._sim.after_step(self)

    # ------------------------------------------------------------------ launch graphs
    GRAPH_STEPS = (16, 2)      # graph sizes (steps per replay), largest first; even -> the AA parity repeats

    def _enqueue_plain_step(self, it):
        """The kernels of one step without output, halo or profiling (what a graph records)."""
        b = self.backend
        kernels = self._kernels_none.primary if (it & 1) == 0 else self._kernels_none.secondary
        for k in kernels:
            b.run_kernel(k, self._regions[1], self._calc_stream)
        base = 1 - (it & 1)
        for axis in self._pbc_axes:
            for k in self._pbc_kernels[base][axis]:
                b.run_kernel(k, None, self._calc_stream)

    def _steps_without_host(self):
        """Number of coming steps that need no host involvement at all: no output, field transfer,
        checkpoint, statistics line, user hook or end-of-run handling."""
        cfg, sim = self.config, self._sim
        it = sim.iteration
        if type(sim).after_step is not LBSim.after_step or sim.need_sync_flag or sim.need_fields_flag or \
                self._checkpoint_req:
            return 0
        lim = [1 << 30]
        if cfg.max_iters > 0:
            lim.append(cfg.max_iters - 1 - it)          # the final step transfers the fields
        if cfg.output_required:
            every = max(1, int(cfg.every))
            first = max(it, int(getattr(cfg, 'from_', 0)))
            nxt = first + ((every - 1 - first) % every)    # first s >= first with (s + 1) % every == 0
            lim.append(nxt - it)
        if getattr(cfg, 'checkpoint_every', 0) > 0 and cfg.checkpoint_file:
            lim.append(cfg.checkpoint_every - 1 - (it % cfg.checkpoint_every))
        if cfg.perf_stats_every > 0:
            lim.append(cfg.perf_stats_every - 1 - (it % cfg.perf_stats_every))
        prof = self._profile
        if prof._is_benchmark and it < prof._sample_from:
            lim.append(prof._sample_from - it)             # sampling starts exactly there
        return max(0, min(lim))

    # ------------------------------------------------------------------ several steps per launch (small 2-D subdomains)
    RESIDENT_STEPS = {'AA': 8, 'AB': 7}       # steps per launch: a halo of 8 nodes either way (csrc/slf_resident.hip)
    RESIDENT_LAUNCHES = (16, 2)               # launches per graph replay, largest first; even: the result is back in the arrays
    RESIDENT_MAX_NODES = 90000                # beyond this a launch per step is faster (measured, see _resident_setup)
    _resident = None

    @staticmethod
    def resident_halo(aa, steps):
        """Halo the tiles of a launch of `steps` steps need, whatever the parity of its first step (the library checks)."""
        return 2 * ((steps + 1) // 2) if aa else steps + 1

    def _resident_setup(self):
        """Kernels, scratch copies and tile shape of the several-steps-per-launch path, or None where it does not apply:
        2-D single-fluid subdomains without neighbours, small enough to be launch-bound, periodic axes wrapped in-sweep,
        node kinds the library's resident kernel serves (it refuses the others)."""
        if self._resident is not None:
            return self._resident or None
        self._resident = False
        cfg, sim, b = self.config, self._sim, self.backend
        if self.dim != 2 or not getattr(cfg, 'hip_resident', True) or os.environ.get('SLF_RESIDENT', '1') == '0' or \
                not hasattr(sim, 'get_resident_kernels') or self._pbc_axes or self.indirect or len(sim.grids) != 1 or \
                self._links or self.has_macro_exchange:
            return None
        d = self._desc
        ext = [(d.lat_nx - 2) if d.periodic_fused[0] else d.lat_nx, (d.lat_ny - 2) if d.periodic_fused[1] else d.lat_ny]
        # Where it pays (256^2 cavity on MI355X, GMLUPS with / without, profiles/r05/ldc2d_resident_variants.txt): in place
        # 17.5 / 13.7, two-copy 16.9 / 12.6, MRT 15.8 / 13.2, 128^2 5.2 / 3.6 -- and where it does not: 512^2 23.8 / 33.6 (the
        # windows hold 4 x the nodes and the launch is bound by instruction issue, so its time grows with the subdomain while
        # a plain sweep of that size is still near its launch-bound plateau), double precision 7.2 / 11.0 (twice the LDS
        # traffic, half the vector rate).  SLF_RESIDENT_FORCE=1 takes the path wherever the kernel applies (the tests).
        if os.environ.get('SLF_RESIDENT_FORCE', '0') != '1':
            if ext[0] * ext[1] > int(os.environ.get('SLF_RESIDENT_MAX_NODES', self.RESIDENT_MAX_NODES)) or \
                    self.float().itemsize != 4:
                return None
        aa = cfg.access_pattern == 'AA'
        steps = int(os.environ.get('SLF_RESIDENT_STEPS', self.RESIDENT_STEPS[cfg.access_pattern]))
        halo = self.resident_halo(aa, steps)
        isz = self.float().itemsize
        # tiles: at most 16 x 16 of them (one workgroup per CU on 256 CUs), windows of at most 2048 nodes in 160 KiB of LDS
        tile = [max(1, -(-e // 16)) for e in ext]
        while (tile[0] + 2 * halo) * (tile[1] + 2 * halo) > 2048 or \
                (tile[0] + 2 * halo) * (tile[1] + 2 * halo) * (9 * isz * 2 + 8) + 4352 > 160 * 1024:   # + the kernel's static tables
            a = 0 if tile[0] >= tile[1] else 1
            if tile[a] == 1:
                return None
            tile[a] -= 1
        nbytes = sim.grid.Q * self._dist_stride * isz
        off = b.dist_align_offset(isz)
        scratch = [b.alloc_buf(size=nbytes, align_offset=off), 0 if aa else b.alloc_buf(size=nbytes, align_offset=off)]
        try:
            fwd, bwd = sim.get_resident_kernels(self, scratch, steps, tile, halo)
        except b.FatalError as e:       # node kinds / formulations the resident kernel does not serve
            cfg.logger.debug('several steps per launch not available: %s' % e)
            for addr in scratch:
                if addr:
                    b.free_buf(addr)
            return None
        pairs = [(self.gpu_dist(0, 0), scratch[0])] + ([] if aa else [(self.gpu_dist(0, 1), scratch[1])])
        self._resident = dict(steps=steps, halo=halo, tile=tile, fwd=fwd, bwd=bwd, pairs=pairs, nbytes=nbytes, graphs={})
        cfg.logger.debug('several steps per launch: %d steps, tiles %s, halo %d' % (steps, tile, halo))
        return self._resident

    def _fast_forward_resident(self, n):
        """As many of the coming n host-free steps as whole graphs of resident launches cover.  Returns the steps done."""
        r = self._resident_setup()
        if r is None or n < 2 * r['steps']:
            return 0
        b, prof, stream = self.backend, self._profile, self._calc_stream
        it = self._sim.iteration
        done = 0
        first = True
        for count in self.RESIDENT_LAUNCHES:
            size = count * r['steps']
            while n - done >= size:
                key = (count, (it + done) & 1)
                if key not in r['graphs']:
                    start = it + done

                    def enqueue():
                        for j in range(count):
                            b.set_iteration(start + j * r['steps'])
                            b.run_kernel(r['fwd'] if (j & 1) == 0 else r['bwd'], None, stream)
                    try:
                        r['graphs'][key] = b.capture_graph(stream, enqueue)
                    except b.FatalError as e:
                        self.config.logger.warning('HIP graph capture of the resident launches failed (%s)' % e)
                        # the path is given up for good: its scratch copies of the arrays and the graphs go with it
                        b.sync_stream(stream)
                        r['graphs'].clear()
                        for _, scratch in r['pairs']:
                            b.free_buf(scratch)
                        self._resident = False
                        b.set_iteration(self._sim.iteration)
                        return done
                if first and not r.get('equalised'):
                    # what no tile covers (padding, the ghost columns of an axis wrapped in-sweep) is the same in both
                    # buffers: no kernel ever writes there, so once is enough -- until the host rewrites the arrays
                    # (_debug_set_dist clears the flag)
                    for dist, scratch in r['pairs']:
                        b.copy_buf_async(scratch, dist, r['nbytes'], stream)
                    r['equalised'] = True
                first = False
                prof.start_step()
                prof.record_gpu_start(TimeProfile.BULK, stream)
                r['graphs'][key].launch(stream)
                prof.record_gpu_end(TimeProfile.BULK, stream, steps=size)
                done += size
                self._sim.iteration = it + done
                prof.end_step(size)
        b.set_iteration(self._sim.iteration)
        return done

    def fast_forward(self):
        """Replays as many of the coming steps as possible as HIP graphs (launch-bound small
        subdomains: one runtime call per 2 / 16 steps instead of a Python-level launch per kernel); small 2-D subdomains
        go through launches that perform several steps each (_fast_forward_resident) first.
        Returns the number of steps done (0: take a normal step)."""
        if not getattr(self.config, 'hip_graphs', True) or self._links or self._quit_requested() or self._time_dependent():
            return 0         # (time-dependent boundary values: the host writes them before every step)
        n = self._steps_without_host()
        if n < 2:
            return 0
        b = self.backend
        done = self._fast_forward_resident(n)
        it = self._sim.iteration - done
        graphs = self.__dict__.setdefault('_graphs', {})
        prof = self._profile
        for size in self.GRAPH_STEPS:
            while n - done >= size:
                key = (size, (it + done) & 1)
                if key not in graphs:
                    start = it + done

                    def enqueue():
                        for s_ in range(size):
                            b.set_iteration(start + s_)
                            self._enqueue_plain_step(start + s_)
                    try:
                        graphs[key] = b.capture_graph(self._calc_stream, enqueue)
                    except b.FatalError as e:
                        # nothing was executed; carry on with plain launches
                        self.config.logger.warning('HIP graph capture failed (%s); continuing without graphs' % e)
                        self.config.hip_graphs = False
                        b.set_iteration(self._sim.iteration)
                        return done
                prof.start_step()
                prof.record_gpu_start(TimeProfile.BULK, self._calc_stream)
                graphs[key].launch(self._calc_stream)
                prof.record_gpu_end(TimeProfile.BULK, self._calc_stream, steps=size)
                done += size
                self._sim.iteration = it + done
                prof.end_step(size)
        b.set_iteration(self._sim.iteration)
        return done

    def _quit_requested(self):
        return self._quit_event is not None and self._quit_event.is_set()

    def _all_streams(self):
        out = [self._calc_stream, self._data_stream]
        bs = getattr(self, '_bnd_stream', None)
        if bs is not None and bs is not self._calc_stream:
            out.append(bs)
        return out

    def finish(self):
        self.backend.sync_stream(*self._all_streams())
        peer = getattr(self._connector, 'peer', None)
        if peer is not None:
            peer.check()        # a wait that gave up (a neighbour that is gone): an error, not wrong numbers
        self.check_gpu_invalid()
        if getattr(self.config, 'final_checkpoint', False) and self.config.checkpoint_file:
            self.save_checkpoint()
        self._sim.after_main_loop(self)
        if self._output is not None:
            self._output.wait()

    def release(self):
        """Gives the device memory of this subdomain back (fields, populations, node map, halo buffers).  The
        runner must not be stepped or queried for device data afterwards; the host copies of the output fields
        (sim.rho, sim.v) stay valid."""
        self.backend.sync_stream(*self._all_streams())
        self._kernels_full = self._kernels_none = self._pbc_kernels = None
        self._links, self._macro_links = {}, {}
        if self._connector is not None:
            self._connector.release(self)
        self.backend.close()

    def main(self):
        """Main loop of a runner that owns its process (one subdomain per GPU process)."""
        cfg = self.config
        t_prev = time.time()
        it_prev = self._sim.iteration
        self._profile.record_start()
        while not self.need_quit():
            if self.fast_forward():
                continue
            sync_req, fields_req, output_req = self.pre_step()
            self._profile.start_step()
            self.step(fields_req)
            self.post_step(sync_req, output_req)
            self._profile.end_step()
            if cfg.perf_stats_every > 0 and self._sim.iteration % cfg.perf_stats_every == 0:
                self.backend.sync_stream(*self._all_streams())
                now = time.time()
                mlups = self.num_fluid_nodes * (self._sim.iteration - it_prev) / (now - t_prev) * 1e-6
                cfg.logger.info('iteration:{0}  speed:{1:.2f} MLUPS'.format(self._sim.iteration, mlups))
                t_prev, it_prev = now, self._sim.iteration
        self.summary = self._profile.record_end()
        self.finish()

    def run(self):
        self.prepare()
        t0 = time.time()
        it0 = self._sim.iteration
        self.main()
        self.timing = {'steps': self._sim.iteration - it0, 'wall': time.time() - t0}


class NNSubdomainRunner(SubdomainRunner):
    """Runner for models with non-local interactions (reference subdomain_runner.py:1840-2197): every
    step first computes the macroscopic fields of all nodes, makes them available on the ghost layer
    (periodic boundaries), and only then collides and streams:

        ShanChenPrepareMacroFields -> ApplyMacroPeriodicBoundaryConditions (rho, phi)
        -> ShanChenCollideAndPropagate0, 1 -> ApplyPeriodicBoundaryConditions (both lattices)

    Axes wrapped inside the kernels need no ghost fill.  Exchange of the macroscopic fields between
    *different* subdomains (reference _send_macro/_recv_macro) is not implemented yet: one subdomain."""

    has_macro_exchange = True

    # -- 1-D decompositions along x of the binary model: the populations of both lattices and the densities cross the
    # -- faces through dense planes the two kernels of a step write / read themselves (xface.NNPlanes)
    _nnx = None

    def _nn_x_faces_only(self):
        if not getattr(self.config, 'hip_xface', True) or self.dim != 3 or len(self._sim.grids) not in (1, 2) or \
                not getattr(self.backend, 'supports_xface_planes', False):
            return False
        if not xface.supported_nn(self._sim.grid, self._desc, self.indirect):
            return False
        # (An edge node in a row next to a y / z face that is not wrapped inside the kernels reads ghost-row entries of the
        # density planes.  They hold what prime() found in the neighbour's ghost row -- the +inf every field is created
        # with outside the lattice, make_scalar_field -- which is what the ghost COLUMN of such a row holds as well: no
        # node owns that position, so build_macro_links never delivers anything there.  A node that computes a force must
        # not sit there in either scheme; a wall does not care.)
        return self._x_slabs_line_up()

    def _init_nn_planes(self):
        """Per neighbour, kind ('dist': both lattices, 'macro': rho and phi) and step parity one send and one receive
        buffer, [through my low face | through my high face] as in _init_xface_halo.  The links carry no pack / unpack
        kernels -- the sweeps fill and read the planes -- so the step program of the general case (pack -> exchange ->
        unpack, per kind) moves them as it stands."""
        spec = self._spec
        nnx = xface.NNPlanes(self.backend, self.module, self._sim.grid, self._desc, n_lat=len(self._sim.grids))
        by_neighbour = {}
        for face, nid in sorted(spec.connecting_subdomains()):
            by_neighbour.setdefault(nid, []).append(xface.LOW if face == spec.X_LOW else xface.HIGH)
        zc = getattr(self._connector, 'zero_copy', False)
        all_links = {'dist': self._links, 'macro': self._macro_links}
        for kind in ('dist', 'macro'):
            n, links = nnx.count[kind], all_links[kind]
            for nid in sorted(by_neighbour):
                link = subdomain_connection.HaloLink(nid)
                link.faces = sorted(by_neighbour[nid])
                link.n_send = link.n_recv = n * len(link.faces)
                if zc:
                    link.recv_bufs = [self._connector.alloc_recv(self, kind, nid, par, link.n_recv, self.float) for par in (0, 1)]
                else:
                    link.send_bufs = [self._connector.alloc_buffer(self, link.n_send, self.float) for _ in (0, 1)]
                    link.recv_bufs = [self._connector.alloc_buffer(self, link.n_recv, self.float) for _ in (0, 1)]
                links[nid] = link
            if zc:
                self._connector.resolve(self)        # collective: once per kind, in this order, on every rank
                for nid, link in links.items():
                    link.send_bufs = [self._connector.send_addr(self, kind, nid, par) for par in (0, 1)]
            for nid, link in links.items():
                link.send_buf, link.recv_buf = link.send_bufs[0], link.recv_bufs[0]
                if kind == 'dist':
                    link.kernels = dict(((mode, copy), ([], [], link.n_send, link.n_recv))
                                        for mode in ('push', 'pull') for copy in (0, 1))
                else:
                    link.packs, link.unpacks = [[], []], [[], []]
        self._nnx = nnx
        nnx.shared = bool(zc)
        self.config.logger.debug('subdomain %d: Shan-Chen model over x-face planes (%s)' % (
            spec.id, 'the neighbours\' memory mapped here' if zc else type(self._connector).__name__))
        for kind in all_links:
            for nid in all_links[kind]:
                self._nnx_place(kind, nid)
        nnx.reset()

    def _nnx_place(self, kind, nid):
        """The face addresses inside the buffers of the link to `nid` (again after a buffer of the link was replaced)."""
        nnx = self._nnx
        link = (self._links if kind == 'dist' else self._macro_links)[nid]
        step = nnx.count[kind] * nnx.isz
        for par in (0, 1):
            for k, face in enumerate(link.faces):                     # my send order: low, high
                nnx.send[kind][par][face] = link.send_bufs[par] + k * step
            for k, face in enumerate(reversed(link.faces)):           # the neighbour's send order seen from here
                nnx.recv[kind][par][face] = link.recv_bufs[par] + k * step

    def _nnx_serial(self, group):
        """Stepped by a group that runs every sweep of its subdomains on ONE stream, planes shared with the neighbours: the
        order of that stream is all the order the planes need (controller.LocalGroup._serialise_sweeps)."""
        return self._nnx is not None and group is not None and getattr(group, 'single_calc_stream', False) and self._nnx.shared

    def halo_messages(self, kind='dist'):
        if self._nnx is None:
            return SubdomainRunner.halo_messages(self, kind)
        par = self._nnx_parity
        links = self._links if kind == 'dist' else self._macro_links
        return [(nid, links[nid].send_bufs[par], links[nid].n_send, links[nid].recv_bufs[par], links[nid].n_recv)
                for nid in sorted(links)]

    def _set_step_state(self, it):
        SubdomainRunner._set_step_state(self, it)
        self._nnx_parity = it & 1

    def _reset_xface(self):
        if self._nnx is None:
            return SubdomainRunner._reset_xface(self)
        self.backend.sync_stream(*self._all_streams())
        self._connector.quiesce(self)       # zero-copy transports: the neighbours write into these planes themselves
        self._nnx.reset(self._calc_stream)
        self.backend.sync_stream(self._calc_stream)
        self._connector.quiesce(self)       # ... and I into theirs: everybody has cleared before anybody fills
        self._nnx_prime()
        group = getattr(self, '_group', None)
        if group is not None and self._nnx.shared:
            # subdomains of one process reset one at a time: what the neighbours had filled in my planes went with the
            # clearing above
            for r in group.runners:
                if r is not self and getattr(r, '_nnx', None) is not None:
                    r._nnx_prime()
        self._connector.quiesce(self)
        self.__dict__.pop('_halo_mode', None)

    def _nnx_prime(self):
        """The density planes I send, from the fields as they are on the device now (xface.NNPlanes.prime)."""
        fields = [self.gpu_field(fp.buffer) for fp in self._sim._scalar_fields if fp.abstract.need_nn]
        self._nnx.prime(fields, self._calc_stream)
        if self.config.access_pattern == 'AA':
            self._nnx.prime_own([self.gpu_dist(g, 0) for g in range(self._nnx.n_lat)], self._calc_stream)
        self.backend.sync_stream(self._calc_stream)

    def _materialise_halo(self):
        if self._nnx is None:
            return SubdomainRunner._materialise_halo(self)
        if not hasattr(self, '_halo_mode'):
            return
        self.backend.sync_stream(*self._all_streams())
        self._nnx.materialise([self.gpu_dist(g, self._halo_copy) for g in range(self._nnx.n_lat)], self._halo_mode == 'push', self._calc_stream,
                              self._nnx_parity)
        self.backend.sync_stream(self._calc_stream)

    def _init_halo(self):
        """Population halo as in the base class (all lattices), plus the exchange of the macroscopic
        fields the non-local force reads at neighbouring nodes (reference _init_interblock_kernels /
        _send_macro / _recv_macro, subdomain_runner.py:1907-2100)."""
        if self._all_specs is not None and len(self._all_specs) >= 2 and self._nn_x_faces_only():
            self._links, self._macro_links, self._ev_halo = {}, {}, None
            if self._connector is None:
                from sailfish_amd.connector import LocalConnector
                self._connector = LocalConnector()
            return self._init_nn_planes()
        SubdomainRunner._init_halo(self)
        if self._all_specs is None or len(self._all_specs) < 2:
            return
        cfg = self.config
        arr = list(reversed(self._physical_size))
        dim = self.dim

        def fused_of(spec):
            return [int(bool(spec._periodicity[a]) and getattr(cfg, 'hip_fused_periodic', True)) for a in range(dim)]

        links = subdomain_connection.build_macro_links(self._spec, self._all_specs, self._global_size,
                                                       self._global_periodic, arr, fused_of)
        b = self.backend
        fields = [self.gpu_field(fp.buffer) for fp in self._sim._scalar_fields if fp.abstract.need_nn]
        isz = np.dtype(self.float).itemsize
        zc = getattr(self._connector, 'zero_copy', False)
        todo = []
        for nid in sorted(links):
            link = links[nid]
            ns, nr = len(link.send), len(link.recv)
            if ns == 0 and nr == 0:
                continue
            link.n_send, link.n_recv = ns * len(fields), nr * len(fields)
            if zc:          # the neighbour's receive buffers mapped here, two sets by step parity (connector.PeerConnector)
                link.recv_bufs = [self._connector.alloc_recv(self, 'macro', nid, par, link.n_recv, self.float) for par in (0, 1)]
            else:
                link.send_buf = self._connector.alloc_buffer(self, link.n_send, self.float)
                link.recv_buf = self._connector.alloc_buffer(self, link.n_recv, self.float)
                link.send_bufs, link.recv_bufs = [link.send_buf] * 2, [link.recv_buf] * 2
            todo.append((nid, link, ns, nr))
        if zc:
            self._connector.resolve(self)
        for nid, link, ns, nr in todo:
            if zc:
                link.send_bufs = [self._connector.send_addr(self, 'macro', nid, par) for par in (0, 1)]
                link.send_buf, link.recv_buf = link.send_bufs[0], link.recv_bufs[0]
            g_s = b.alloc_buf(like=link.send) if ns else 0
            g_r = b.alloc_buf(like=link.recv) if nr else 0
            # [parity of the step] -> kernels
            link.packs = [[self.get_kernel('CollectSparseData', [g_s, f, link.send_bufs[par] + i * ns * isz, ns], 'PPPi')
                           for i, f in enumerate(fields)] if ns else [] for par in (0, 1)]
            link.unpacks = [[self.get_kernel('DistributeSparseData', [g_r, f, link.recv_bufs[par] + i * nr * isz, nr], 'PPPi')
                             for i, f in enumerate(fields)] if nr else [] for par in (0, 1)]
            self._macro_links[nid] = link

    def _prepare_compute_kernels(self):
        self._kernels_full = self._sim.get_compute_kernels(self, True, True)
        self._kernels_none = self._sim.get_compute_kernels(self, False, True)
        self._pbc_kernels = self._sim.get_pbc_kernels(self)
        self._pbc_axes = [a for a in range(self.dim) if self._local_periodic()[a] and not self._fused[a]]
        self._regions = self._make_regions()
        self._kernels_prepared = True

    def _enqueue_plain_step(self, it):
        b = self.backend
        macro_kernel, sim_kernels = self._kernels_none[it & 1]
        base = 1 - (it & 1)
        b.run_kernel(macro_kernel, None, self._calc_stream)
        for axis in self._pbc_axes:
            for k in self._pbc_kernels.macro[base][axis]:
                b.run_kernel(k, None, self._calc_stream)
        for k in sim_kernels:
            b.run_kernel(k, None, self._calc_stream)
        for axis in self._pbc_axes:
            for k in self._pbc_kernels.distributions[base][axis]:
                b.run_kernel(k, None, self._calc_stream)

    def _dist_pbc_kernels(self):
        return self._pbc_kernels.distributions

    def _sweep_kernels(self, it, sync_req):
        return (self._kernels_full if sync_req else self._kernels_none)[it & 1][1]

    def _program_macro(self, q, it, group=None, sync_req=False):
        """Macroscopic fields of every real node, their local periodic images, and the exchange of the values the
        non-local force reads in the neighbours' territory (reference NNSubdomainRunner.step, subdomain_runner.py:
        2102-2197): calc stream -> event -> data stream: pack -> [exchange] -> unpack -> event -> calc stream."""
        prof, timed = self._profile, not q.planned
        ev, pev = self._pev[it & 1], self._pev[1 - (it & 1)]
        sk = self._calc_stream
        # the kernels of an output step store every field the host reads afterwards (binary Shan-Chen: the velocity,
        # which the other steps' pass leaves to the sweep -- lb_binary.get_compute_kernels)
        macro_kernel = (self._kernels_full if sync_req else self._kernels_none)[it & 1][0]
        base = 1 - (it & 1)
        if self._nnx is not None:
            self._nnx.program_bind(q, it)
            if self._nnx_serial(group):
                q.launch(macro_kernel, None, sk)
                return
        if self._links:
            q.wait(sk, pev['halo'])                          # populations received after the last step
        if timed:
            prof.record_gpu_start(TimeProfile.MACRO_BULK, sk)
        q.launch(macro_kernel, None, sk)
        if timed:
            prof.record_gpu_end(TimeProfile.MACRO_BULK, sk)
        for axis in self._pbc_axes:
            for k in self._pbc_kernels.macro[base][axis]:
                q.launch(k, None, sk)
        if self._macro_links:
            q.record(ev['macro'], sk)
            self._program_pack(q, it, ev['macro'], group, kind='macro')
            if group is None:
                self._program_macro_back(q, it)

    def _program_macro_back(self, q, it):
        """Unpack the neighbours' values into the ghost nodes; the sweep waits for it."""
        if not self._macro_links or self._nnx_serial(getattr(self, '_group', None)):
            return
        prof, timed = self._profile, not q.planned
        ev, sh = self._pev[it & 1], self._data_stream
        if timed:
            prof.record_gpu_start(TimeProfile.MACRO_DISTRIB, sh)
        for nid in sorted(self._macro_links):
            for k in self._macro_links[nid].unpacks[it & 1]:
                q.launch(k, None, sh)
        if timed:
            prof.record_gpu_end(TimeProfile.MACRO_DISTRIB, sh)
        q.record(ev['macro_halo'], sh)
        q.wait(self._calc_stream, ev['macro_halo'])

    def _program_front(self, q, it, sync_req, group=None):
        """The sweeps of every lattice over the whole subdomain (no face layers split off: the force reads the fields of
        the step, which the macro pass has just written) -> ghost-layer PBC kernels -> pack."""
        if self._nnx_serial(group):
            for k in self._sweep_kernels(it, sync_req):
                q.launch(k, None, self._calc_stream)
            return
        self._program_sweep_rest(q, it, self._sweep_kernels(it, sync_req), None, None, group)

    def _program_back(self, q, it):
        if self._nnx_serial(getattr(self, '_group', None)):
            return
        SubdomainRunner._program_back(self, q, it)

    def _debug_get_dist(self, output=True, grid_num=0, copy=None):
        return SubdomainRunner._debug_get_dist(self, output, grid_num, copy)
