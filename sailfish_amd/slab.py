"""Periodic box cut into slabs along one axis, one slab per GPU (the scaling benchmark geometry: SURVEY.md §8(d)
M0 / C4, reference geo.py:100-135 EqualSubdomainsGeometry3D with conn_axis = x | y | z).

Per step (reference subdomain_runner.py:1028-1058 boundary / bulk split):
  calc stream : wait(previous halo) -> sweep the two face layers -> event -> sweep the interior
  halo stream : wait(event) -> pack the two faces -> RCCL send / recv with the two ring neighbours -> unpack -> event
The other two axes are wrapped inside the sweep; the split axis is wrapped inside the sweep when there is a single
slab, otherwise it goes through the ghost layers + halo.  x faces cannot be split off (a workgroup owns whole rows):
with axis = 'x' the sweep is cut into z-chunks instead, and the planes of the face buffers a chunk has completed
travel while the next chunk computes (xface.ChunkPlan); a chunk of the next step waits only for the planes it reads.

z and y faces are packed with the box kernels (Collect / DistributeContinuousData: no index lists; reference
kernel_utils.mako:526-543, 629-645) and move contiguous row segments.  x faces are not packed at all: the edge lanes
of the sweep write / read dense face buffers (xface.py), the halo stream only exchanges them; the box kernels remain
the fallback (SLF_XFACE=0, rows longer than 1024 nodes) and then gather with stride arr_nx.

Which layer travels (reference subdomain_runner.py:1069-1103, Appendix A.4-5 of SURVEY.md):
  push steps (AB, odd AA): my ghost layer (populations pushed across the face) -> neighbour's first real layer, same slots;
  even AA step (in-place, opposite slots): my first real layer, opposite slots -> neighbour's ghost layer, so that
      its next (odd) step can pull them.
"""
import os

import numpy as np

from sailfish_amd import hipabi, sym, xface
from sailfish_amd.backend_hip import DirectQueue, HIPEvent, NotPlannable
from sailfish_amd.box import BoxSim, make_box_desc

AXES = {'x': 0, 'y': 1, 'z': 2}


class SlabPlan(object):
    """Pure host logic: the node boxes (base, strides, extents) and direction masks of the halo pack / unpack."""

    def __init__(self, grid, desc, axis=2):
        self.grid, self.desc, self.axis = grid, desc, axis
        self.n = [desc.lat_nx - 2, desc.lat_ny - 2, desc.lat_nz - 2]
        self.up_dists = sym.get_prop_dists(grid, 1, axis)       # e_axis = +1
        self.down_dists = sym.get_prop_dists(grid, -1, axis)    # e_axis = -1
        others = [a for a in range(3) if a != axis]
        self.col_axis, self.row_axis = others                   # columns = the faster-varying of the two
        self.ncols, self.nrows = self.n[self.col_axis], self.n[self.row_axis]
        self.count = len(self.up_dists) * self.ncols * self.nrows
        self.strides = [1, desc.arr_nx, desc.arr_nx * desc.arr_ny]

    def box(self, layer, dists):
        """(dirs mask, base node, col stride, ncols, row stride, nrows) of the face layer at coordinate `layer`."""
        coord = [1, 1, 1]
        coord[self.axis] = layer
        base = sum(c * s for c, s in zip(coord, self.strides))
        mask = 0
        for q in dists:
            mask |= 1 << q
        return (mask, base, self.strides[self.col_axis], self.ncols, self.strides[self.row_axis], self.nrows)

    def boxes(self, swap):
        """(send_up, send_down, recv_low, recv_high); swap = True after the even AA step."""
        n = self.n[self.axis]
        opp = self.grid.idx_opposite
        if not swap:
            return (self.box(n + 1, self.up_dists), self.box(0, self.down_dists),
                    self.box(1, self.up_dists), self.box(n, self.down_dists))
        up_s = [opp[i] for i in self.up_dists]
        dn_s = [opp[i] for i in self.down_dists]
        return (self.box(n, up_s), self.box(1, dn_s), self.box(0, up_s), self.box(n + 1, dn_s))

    def index_list(self, box):
        """The same box as a q * dist_size + node index list, buffer order [k][row][col] (tests, oracle twin)."""
        mask, base, cs, nc, rs, nr = box
        ds = hipabi.dist_stride(self.desc)
        r, c = np.meshgrid(np.arange(nr, dtype=np.uint64), np.arange(nc, dtype=np.uint64), indexing='ij')
        node = (np.uint64(base) + c * np.uint64(cs) + r * np.uint64(rs)).ravel()
        qs = [q for q in range(self.grid.Q) if mask >> q & 1]
        return np.concatenate([np.uint64(q) * np.uint64(ds) + node for q in qs])

    def regions(self):
        """(boundary regions, bulk region) as (y0, y1, z0, z1) row ranges; no split for x faces / thin slabs."""
        nx, ny, nz = self.n
        full = (1, ny + 1, 1, nz + 1)
        if self.axis == 2 and nz > 2:
            return [(1, ny + 1, 1, 2), (1, ny + 1, nz, nz + 1)], (1, ny + 1, 2, nz)
        if self.axis == 1 and ny > 2:
            return [(1, 2, 1, nz + 1), (ny, ny + 1, 1, nz + 1)], (2, ny, 1, nz + 1)
        return [], full


class SlabSim(BoxSim):
    def __init__(self, backend, grid, size, rank=0, world=1, model='bgk', precision='single',
                 access_pattern='AA', visc=1.0 / 6.0, fused_periodic=True, exchanger=None, axis='z',
                 force_halo=False, tune_placement=False):
        """size: the LOCAL slab (nx, ny, nz); the global box is `world` slabs stacked along `axis`.
        tune_placement: place the distribution arrays by measurement (placement.choose)."""
        self.grid, self.size, self.rank, self.world = grid, size, rank, world
        self.axis = AXES[axis] if isinstance(axis, str) else int(axis)
        assert grid.dim == 3
        self.halo = world > 1 or force_halo
        if self.halo and not fused_periodic:
            raise ValueError('multi-slab runs wrap the unsplit axes inside the sweep')
        fused = [int(fused_periodic)] * 3
        if self.halo:
            fused[self.axis] = 0
        desc = make_box_desc(grid, size, model=model, precision=precision, access_pattern=access_pattern, visc=visc,
                             periodic_fused=fused, fluid_only=True)
        periodic = [True, True, True]
        periodic[self.axis] = not self.halo
        BoxSim.__init__(self, backend, desc, periodic=tuple(periodic), tune_placement=tune_placement)
        self.calc_stream = self.stream
        self.block_size = self.module.block_size
        self.halo_ms = []
        if self.halo:
            self._init_halo(exchanger)

    # -- halo machinery ------------------------------------------------------
    def _init_halo(self, exchanger):
        import torch
        from sailfish_amd.connector import init_distributed, make_ring_exchanger
        from sailfish_amd import peer as peer_mod
        b = self.backend
        self.peer = self.peer_group = None
        if exchanger is None:
            init_distributed()
            # the neighbours' receive buffers mapped into this process (sailfish_amd/peer.py): what the sweep's edge lanes
            # / the pack kernels write IS what the neighbour reads, the exchange is a pair of counters.  None where the
            # ranks cannot map each other's memory (or SLF_HALO_TRANSPORT says otherwise): RCCL / torch.distributed then
            self.peer = peer_mod.process_transport(b, self.rank, self.world)
        self.plan = SlabPlan(self.grid, self.desc, self.axis)
        self.exchanger = exchanger or (None if self.peer is not None else make_ring_exchanger(self.rank, self.world, b))
        self.neighbours = sorted(set([(self.rank + 1) % self.world, (self.rank - 1) % self.world]))
        self._peer_open = None     # x faces over the peer transport: (kind of the last step, kind of the next) while its signals are unconsumed
        self.halo_stream = b.make_stream(high_priority=os.environ.get('SLF_HALO_PRIORITY', '1') != '0')
        self.t_halo_stream = torch.cuda.ExternalStream(self.halo_stream.native, device=torch.device('cuda', b.gpu_id))
        # a second calc stream: z / y slabs sweep their face layers on it, so that the stream of the interior sweep never
        # waits for a halo; x slabs alternate their z-chunks between the two, so that the chip does not drain at every
        # chunk boundary (SLF_CALC_STREAMS=1: everything on one stream, the round-3 scheme)
        self.calc_stream2 = b.make_stream() if os.environ.get('SLF_CALC_STREAMS', '2') != '1' else self.stream
        # an exchanger that can be called in the middle of a step (RCCL / gloo / ring of one): the step is one program
        # (_program), replayed from a C-ABI step plan where the transport allows; exchangers that move whole buffers
        # between step_compute() and step_finish() (tests: two slabs in one process) keep the three-phase protocol
        self.mid_step = hasattr(self.exchanger, 'exchange_range') or self.peer is not None
        self._plans = {}
        self._plan_ok = getattr(b, 'supports_step_plans', False) and os.environ.get('SLF_STEP_PLAN', '1') != '0'
        tdtype = torch.float32 if self.desc.precision == 4 else torch.float64
        dev = torch.device('cuda', b.gpu_id)
        self.xface = None
        self.ev_halo = None
        self.regs_bnd, self.reg_bulk = self.plan.regions()
        self.time_halo = False
        if self.axis == 0 and xface.supported(self.grid, self.desc) and os.environ.get('SLF_XFACE', '1') != '0':
            # x faces: the sweep's edge lanes write / read dense face buffers themselves (xface.py) -- no pack / unpack
            tensors = []

            def alloc(n):
                tensors.append(torch.empty(n, dtype=tdtype, device=dev))
                return tensors[-1].data_ptr()
            if self.peer is not None:
                # my send planes ARE the neighbours' receive planes: what leaves through my low face is what the rank
                # below receives through its high face
                recv = self._peer_buffers(xface.face_count(self.desc))
                up, down = (self.rank + 1) % self.world, (self.rank - 1) % self.world
                send = [[self.peer_group.lookup(down, ('recv', par, xface.HIGH)),
                         self.peer_group.lookup(up, ('recv', par, xface.LOW))] for par in (0, 1)]
                self.xface = xface.XFaceHalo(b, self.module, self.grid, self.desc, send, recv, shared=True)
                self.t_sets = self.t_bufs = None
            else:
                self.xface = xface.XFaceHalo.allocate(b, self.module, self.grid, self.desc, (True, True), alloc)
                # per parity: send low, send high, receive low, receive high  ->  s_up s_down r_low r_high
                self.t_sets = [[tensors[4 * p + 1], tensors[4 * p + 0], tensors[4 * p + 2], tensors[4 * p + 3]] for p in (0, 1)]
                self.t_bufs = self.t_sets[0]
            self._reset_faces()
            # overlap = every batch of planes is exchanged as soon as its chunks are done
            self.overlap = self.mid_step
            self.chunks = xface.ChunkPlan(self.size[2], self.desc.periodic_fused[2], None if self.overlap else 1)
            self._batch_events, self._prev_kind = None, None
            self._event_pool = {}
            n = len(self.chunks.order)
            self._ev_chunk = [[HIPEvent(b) for _ in range(n)] for _ in (0, 1)]
            self._ev_batch = [[HIPEvent(b) for _ in range(n)] for _ in (0, 1)]
            return
        n = self._count = self.plan.count
        if self.peer is not None:
            # the pack kernels write the neighbours' receive buffers themselves; two sets that alternate by step parity,
            # so that a set is written again only after the neighbour's unpack of two steps ago -- which precedes, on its
            # halo stream, the signal of the step in between that this rank has waited for by then
            recv = self._peer_buffers(n)
            up, down = (self.rank + 1) % self.world, (self.rank - 1) % self.world
            bufs = [[self.peer_group.lookup(up, ('recv', par, 0)), self.peer_group.lookup(down, ('recv', par, 1)),
                     recv[par][0], recv[par][1]] for par in (0, 1)]
            self.t_bufs = None
        else:
            self.t_bufs = [torch.empty(n, dtype=tdtype, device=dev) for _ in range(4)]  # s_up s_down r_low r_high
            bufs = [[t.data_ptr() for t in self.t_bufs]] * 2
        self.k_halo = {}
        for swap in ((False, True) if self.aa else (False,)):
            boxes = self.plan.boxes(swap)
            for di, dbuf in enumerate(self.gpu_dist):
                # the step that uses these kernels: in place the even ones (swap), two-copy the ones that write copy di
                par = (0 if swap else 1) if self.aa else 1 - di
                ks = []
                for j in range(4):
                    name = 'CollectContinuousData' if j < 2 else 'DistributeContinuousData'
                    ks.append(b.get_kernel(self.module, name, (64,), [dbuf, bufs[par][j]] + list(boxes[j]), 'PPiiiiii'))
                self.k_halo[(swap, di)] = ks
        self._ev = [dict((name, HIPEvent(b)) for name in ('bnd', 'bulk', 'halo')) for _ in (0, 1)]

    def _peer_buffers(self, count):
        """[parity][low, high] receive buffers of `count` reals other processes can map, published under
        ('recv', parity, face): collective, every rank in the same order."""
        isz = 4 if self.desc.precision == 4 else 8
        grp = self.peer_group = self.peer.group()
        recv = [[grp.alloc(count * isz) for _ in (0, 1)] for _ in (0, 1)]
        for par in (0, 1):
            for a in recv[par]:
                self.backend.memset_buf(a, 0xFF, count * isz)
        self.backend.sync()
        grp.publish(dict((('recv', par, f), recv[par][f]) for par in (0, 1) for f in (0, 1)))
        return recv

    def _reset_faces(self):
        """x-face buffers: nothing has crossed the faces (initial state, a state written from the host).  With the peer
        transport the neighbours write into my receive planes: nobody is stepping while they are filled."""
        if self.peer is not None:
            self.sync()
            self.peer.barrier()
        self.xface.reset(self.stream)
        self._batch_events, self._prev_kind = None, None
        if self.peer is not None:
            self.sync()
            self.peer.barrier()

    # -- one step as a program: written once against the DirectQueue / HIPPlan interface (backend_hip.py) -------------
    def _exchange(self, q, ranges):
        """One group of transfers on the halo stream: the element ranges [(first, count)] of the four face buffers
        (send up, send down, receive low, receive high)."""
        if self.peer is not None:
            # nothing to move: what the halo stream has waited for so far is in the neighbours' memory already
            from sailfish_amd.peer import CH_DIST
            q.peer_signal(self.peer, self.neighbours, CH_DIST, self.halo_stream)
            q.peer_wait(self.peer, self.neighbours, CH_DIST, self.halo_stream)
            return
        ex = self.exchanger
        bufs = self.t_bufs
        if getattr(ex, 'direct', False):                    # straight to RCCL (connector.RcclRingExchanger)
            q.exchange(ex.rccl, ex.batch(bufs, ranges), self.halo_stream)
            return
        if ex.plain_copy():                                 # a ring of one without a process group
            isz = bufs[0].element_size()
            for first, count in ranges:
                for src, dst in ((bufs[0], bufs[2]), (bufs[1], bufs[3])):
                    q.copy(dst.data_ptr() + first * isz, src.data_ptr() + first * isz, count * isz, self.halo_stream)
            return

        def run():                                          # torch.distributed (gloo in the tests): needs Python
            import torch
            with torch.cuda.stream(self.t_halo_stream):
                for first, count in ranges:
                    ex.exchange_range(bufs, first, count)
        q.call(run)

    def _program(self, q, it, save_macro):
        if self.xface is not None:
            if self.peer is not None:
                return self._program_xface_peer(q, it, save_macro)
            return self._program_xface(q, it, save_macro)
        return self._program_box(q, it, save_macro)

    def _step_kinds(self, it):
        kind = 'own' if (self.aa and (it & 1) == 0) else 'push'
        other = 'push' if (not self.aa or kind == 'own') else 'own'        # the step before and the step after
        return kind, other

    def _program_xface_peer(self, q, it, save_macro):
        """x slabs whose send planes ARE the neighbours' receive planes (peer transport).  Calc stream: [wait] chunk
        [wait] chunk ...; halo stream: after every chunk the neighbours' next step counts on, a signal.  A wait is a
        one-lane kernel IN FRONT OF THE CHUNK THAT NEEDS IT: it asks for the signals of the neighbours' previous step up
        to the position after which the planes the chunk reads are complete and the planes it writes have been read
        (xface.ChunkPlan.peer_counts) -- by then they have normally arrived, and a neighbour that is late holds back
        exactly the work that depends on it (a wait enqueued right behind this step's own signal, on a stream of its
        own, spins until the neighbours are as far as this rank -- and stalls whatever shares its hardware queue: eight
        processes on one device ran at 7-28 instead of 38 GMLUPS that way, profiles/r06/peer_eight_processes.txt).
        The first step after the counters were drained (self._peer_open is None) has nothing to wait for."""
        from sailfish_amd.peer import CH_DIST
        x, plan, ny = self.xface, self.chunks, self.size[1]
        k = self.k_sweep[int(save_macro)][0] if self.aa else self.k_sweep[int(save_macro)][it & 1]
        kind, other = self._step_kinds(it)
        par = it & 1
        snd, rcv = x.send[par], x.recv[1 - par]
        q.xface(self.module, snd[xface.LOW], snd[xface.HIGH], rcv[xface.LOW], rcv[xface.HIGH])
        sk, sh = self.calc_stream, self.halo_stream
        evc = self._ev_chunk[par]
        waits = dict(plan.peer_counts(kind, other)) if self._peer_open is not None else {}
        signals = set(plan.peer_signals(kind, other))
        t0 = None
        for pos, c in enumerate(plan.order):
            if pos in waits:
                q.peer_wait(self.peer, self.neighbours, CH_DIST, sk, waits[pos])
            q.launch(k, plan.region(c, ny), sk)
            if pos not in signals:
                continue
            q.record(evc[pos], sk)
            q.wait(sh, evc[pos])
            if self.time_halo and not q.planned and t0 is None:
                t0 = self.backend.make_event(sh, timing=True)
            q.peer_signal(self.peer, self.neighbours, CH_DIST, sh)
        if t0 is not None:
            self._halo_events.append((t0, self.backend.make_event(sh, timing=True)))

    def _peer_drain(self):
        """Consumes the signals of the neighbours' last step that no step of this rank has waited for yet (the waits of
        a step refer to the step before): afterwards the counters of both sides agree again and the next step starts
        as a first step.  Before anything rewrites the state or lets go of the buffers."""
        if self.peer is None or getattr(self, '_peer_open', None) is None:
            return
        if self.xface is None:
            self._box_back(DirectQueue(self.backend), self.iteration - 1)
        else:
            from sailfish_amd.peer import CH_DIST
            kind, nxt = self._peer_open
            n = len(self.chunks.peer_signals(kind, nxt))
            if n:
                self.peer.wait(self.neighbours, CH_DIST, self.calc_stream, n)
        self._peer_open = None

    def _program_box(self, q, it, save_macro):
        """z / y slabs.  Boundary stream: wait(previous halo, previous interior) -> the two face layers -> event;
        interior stream: wait(previous face layers) -> interior; halo stream: wait(face layers) -> pack -> exchange ->
        unpack -> event.  The interior sweep depends on the neighbours only through the face layers of the step before
        (what the unpack writes -- the first real layer after a push step, the ghost layer after the even in-place
        step -- is read by the face layers alone), so its stream never waits for a transfer."""
        if self.aa:
            k, out, swap = self.k_sweep[int(save_macro)][0], 0, (it & 1) == 0
        else:
            k, out, swap = self.k_sweep[int(save_macro)][it & 1], 1 - (it & 1), False
        ev, pev = self._ev[it & 1], self._ev[1 - (it & 1)]
        sk, sb, sh = self.calc_stream, self.calc_stream2, self.halo_stream
        if self.peer is not None and self._peer_open is not None:
            self._box_back(q, it - 1)
        if self.regs_bnd:
            q.wait(sb, pev['halo'])
            if sb is not sk:
                q.wait(sb, pev['bulk'])
            for reg in self.regs_bnd:
                q.launch(k, reg, sb)
            q.record(ev['bnd'], sb)
            if sb is not sk:
                q.wait(sk, pev['bnd'])
            q.launch(k, self.reg_bulk, sk)
            q.record(ev['bulk'], sk)
        else:       # a layer that cannot be split off: the whole sweep first
            q.wait(sk, pev['halo'])
            q.launch(k, self.reg_bulk, sk)
            q.record(ev['bnd'], sk)
        q.wait(sh, ev['bnd'])
        ks = self.k_halo[(swap, out)]
        t0 = self.backend.make_event(sh, timing=True) if (self.time_halo and not q.planned) else None
        q.launch(ks[0], None, sh)
        q.launch(ks[1], None, sh)
        if self.peer is not None:
            # the pack kernels have written the neighbours' receive buffers: tell them.  The other half of the exchange --
            # wait for THEIR signal, unpack -- is enqueued in front of the next step (_box_back): by then the signal has
            # normally arrived, and a wait that does spin holds back the face layers that depend on it, nothing else
            from sailfish_amd.peer import CH_DIST
            q.peer_signal(self.peer, self.neighbours, CH_DIST, sh)
        else:
            self._exchange(q, [(0, self._count)])
            q.launch(ks[2], None, sh)
            q.launch(ks[3], None, sh)
            q.record(ev['halo'], sh)
        if t0 is not None:
            self._halo_events.append((t0, self.backend.make_event(sh, timing=True)))

    def _box_back(self, q, it):
        """Peer transport, z / y slabs: the second half of step `it`'s exchange -- wait for the neighbours' signal of that
        step, unpack what they wrote, record the event the next face layers wait for."""
        from sailfish_amd.peer import CH_DIST
        if self.aa:
            out, swap = 0, (it & 1) == 0
        else:
            out, swap = 1 - (it & 1), False
        ks, sh = self.k_halo[(swap, out)], self.halo_stream
        q.peer_wait(self.peer, self.neighbours, CH_DIST, sh, 1)
        q.launch(ks[2], None, sh)
        q.launch(ks[3], None, sh)
        q.record(self._ev[it & 1]['halo'], sh)

    def _program_xface(self, q, it, save_macro):
        """x slabs: the sweep in z-chunks, alternating between the two calc streams (a chunk starts while the one
        before it drains); after each chunk the planes of the send buffers it completed travel on the halo stream.  A
        chunk waits for (a) the transfers of the previous step that carry the planes it reads and (b) the chunks of the
        previous step that touched its planes or their neighbours and ran on the other stream."""
        x, plan, ny = self.xface, self.chunks, self.size[1]
        k = self.k_sweep[int(save_macro)][0] if self.aa else self.k_sweep[int(save_macro)][it & 1]
        kind = 'own' if (self.aa and (it & 1) == 0) else 'push'
        prev_kind = 'push' if (not self.aa or kind == 'own') else 'own'
        par = it & 1
        if self.t_sets is not None:
            self.t_bufs = self.t_sets[par]
        # the z-chunks alternate between the two calc streams only on request (SLF_XFACE_STREAMS=2): a chunk that starts
        # while the one before it drains was measured SLOWER than the drain it avoids (profiles/r04/NOTES.md)
        streams = [self.calc_stream, self.calc_stream2 if os.environ.get('SLF_XFACE_STREAMS', '1') == '2' else self.calc_stream]
        if x.needs_clear:
            streams[1] = streams[0]
        snd, rcv = x.send[par], x.recv[1 - par]
        q.xface(self.module, snd[xface.LOW], snd[xface.HIGH], rcv[xface.LOW], rcv[xface.HIGH])
        if x.needs_clear:
            for a in snd:
                if a:
                    q.memset(a, 0xFF, x.nbytes, streams[0])
        evc, evb = self._ev_chunk[par], self._ev_batch[par]
        pevc, pevb = self._ev_chunk[1 - par], self._ev_batch[1 - par]
        # buffers that are not copied (peer transport): a chunk also waits until the planes it writes have been read
        need = plan.peer_need(kind, prev_kind) if x.shared else plan.need[prev_kind]
        pos_of = dict((c, pos) for pos, c in enumerate(plan.order))
        sh = self.halo_stream
        t0 = None
        waited = {}
        for pos, c in enumerate(plan.order):
            st = streams[pos & 1]
            if need[c] > waited.get(id(st), -1):     # the streams are in order: a later transfer waited for covers the earlier ones
                q.wait(st, pevb[need[c]])
                waited[id(st)] = need[c]
            for c2 in plan.neighbours(c):
                if streams[pos_of[c2] & 1] is not st:
                    q.wait(st, pevc[pos_of[c2]])
            q.launch(k, plan.region(c, ny), st)
            if not plan.exchanges_at(pos) and streams[0] is streams[1] and not x.shared:
                continue                 # nothing travels after this chunk and nobody waits for it
            q.record(evc[pos], st)
            q.wait(sh, evc[pos])
            if self.time_halo and not q.planned and t0 is None:
                t0 = self.backend.make_event(sh, timing=True)
            runs = plan.batches[kind][pos]
            if runs or x.shared:
                self._exchange(q, [(p0 * x.plane, (p1 - p0) * x.plane) for p0, p1 in runs])
            q.record(evb[pos], sh)
        if t0 is not None:
            self._halo_events.append((t0, self.backend.make_event(sh, timing=True)))

    def step_compute(self, save_macro=False):
        """Three-phase protocol (exchangers that move whole buffers between step_compute() and step_finish()): face
        layers, event, interior on the calc stream; halo pack on the halo stream."""
        self.backend.set_iteration(self.iteration)
        if self.xface is not None:
            return self._step_compute_xface(save_macro)
        b = self.backend
        it = self.iteration
        if self.aa:
            k, out, swap = self.k_sweep[int(save_macro)][0], 0, (it & 1) == 0
        else:
            k, out, swap = self.k_sweep[int(save_macro)][it & 1], 1 - (it & 1), False
        if self.ev_halo is not None:
            self.calc_stream.wait_for_event(self.ev_halo)
        for reg in self.regs_bnd:
            b.run_kernel(k, reg, self.calc_stream)
        if self.regs_bnd:
            ev_bnd = b.make_event(self.calc_stream)
            b.run_kernel(k, self.reg_bulk, self.calc_stream)
        else:
            b.run_kernel(k, self.reg_bulk, self.calc_stream)
            ev_bnd = b.make_event(self.calc_stream)
        self.halo_stream.wait_for_event(ev_bnd)
        self._ks = self.k_halo[(swap, out)]
        b.run_kernel(self._ks[0], None, self.halo_stream)
        b.run_kernel(self._ks[1], None, self.halo_stream)
        self.iteration += 1

    def _step_compute_xface(self, save_macro):
        """x-slabs, three-phase protocol: one launch of the whole sweep (ChunkPlan with a single chunk); the caller
        moves the face buffers between step_compute() and step_finish()."""
        b = self.backend
        it = self.iteration
        k = self.k_sweep[int(save_macro)][0] if self.aa else self.k_sweep[int(save_macro)][it & 1]
        plan, ny = self.chunks, self.size[1]
        par = self.xface.begin_step(it, self.calc_stream)
        self.t_bufs = self.t_sets[par]
        prev = self._batch_events
        for pos, c in enumerate(plan.order):
            if prev is not None:
                self.calc_stream.wait_for_event(prev[pos])
            b.run_kernel(k, plan.region(c, ny), self.calc_stream)
            self.halo_stream.wait_for_event(b.make_event(self.calc_stream))
        self.iteration += 1

    def step_exchange(self):
        import torch
        with torch.cuda.stream(self.t_halo_stream):
            self.exchanger.exchange(*self.t_bufs)

    def step_finish(self):
        b = self.backend
        if self.xface is None:
            b.run_kernel(self._ks[2], None, self.halo_stream)
            b.run_kernel(self._ks[3], None, self.halo_stream)
            self.ev_halo = b.make_event(self.halo_stream)
            return
        ev = b.make_event(self.halo_stream)      # whole buffers were moved by the caller: one event for every chunk
        self._batch_events = [ev] * len(self.chunks.order)

    def step(self, save_macro=False, region=None):
        if not self.halo:
            return BoxSim.step(self, save_macro)
        if not self.mid_step:
            self.step_compute(save_macro)
            self.step_exchange()
            self.step_finish()
            return
        b = self.backend
        it = self.iteration
        peer_x = self.peer is not None
        first = peer_x and self._peer_open is None      # nothing of a previous step to wait for: not the plan's program
        if self._plan_ok and not self.time_halo and not first:
            key = (it & 1, int(bool(save_macro)))
            plan = self._plans.get(key)
            if plan is None:
                plan = b.make_plan()
                try:
                    self._program(plan, it, save_macro)
                    self._plans[key] = plan
                except NotPlannable:            # the transport needs Python between the launches
                    self._plan_ok, plan = False, None
            if plan is not None:
                if self.xface is not None and self.t_sets is not None:
                    self.t_bufs = self.t_sets[it & 1]
                plan.run(it)
                self.iteration += 1
                b.set_iteration(self.iteration)     # kernels launched through run_kernel() next see the new parity
                if self.xface is not None:
                    self.xface._bound = None        # the plan set the module's face buffers itself
                if peer_x:
                    self._peer_open = self._step_kinds(it)
                return
        b.set_iteration(it)
        self._program(DirectQueue(b), it, save_macro)
        self.iteration += 1
        b.set_iteration(self.iteration)
        if self.xface is not None:
            self.xface._bound = None
        if peer_x:
            self._peer_open = self._step_kinds(it)      # (kind of the step just enqueued, kind of the one that follows)

    def release(self):
        BoxSim.release(self)            # (sync() drains the peer counters)
        if getattr(self, 'peer_group', None) is not None:
            self.peer_group.release()       # collective: every rank lets go of its simulation at the same point
            self.peer_group = None

    def step_sweep_only(self):
        """The sweep launches of one step without any halo traffic (timing reference: what the calc stream costs
        when nothing has to be waited for).  Leaves the slab faces stale -- re-initialise afterwards."""
        b = self.backend
        it = self.iteration
        b.set_iteration(it)
        k = self.k_sweep[0][0] if self.aa else self.k_sweep[0][it & 1]
        if self.halo and self.xface is not None:
            self.xface.begin_step(it, self.calc_stream)
            for c in self.chunks.order:
                b.run_kernel(k, self.chunks.region(c, self.size[1]), self.calc_stream)
        else:
            for reg in getattr(self, 'regs_bnd', []):
                b.run_kernel(k, reg, self.calc_stream)
            b.run_kernel(k, getattr(self, 'reg_bulk', None), self.calc_stream)
        self.iteration += 1
        b.set_iteration(self.iteration)

    def start_halo_timing(self):
        self.time_halo, self._halo_events = True, []

    def stop_halo_timing(self):
        """Mean time (ms) the halo stream was busy per step (pack -> exchange -> unpack) since start_halo_timing()."""
        self.time_halo = False
        self.sync()
        ms = [e1.time_since(e0) for e0, e1 in self._halo_events]
        self._halo_events = []
        return float(np.mean(ms)) if ms else 0.0

    def sync(self):
        """Everything this slab has enqueued is done -- and, with the peer transport, everything its neighbours' last
        step wrote into its face buffers (the waits of a step refer to the step before: the last step's are due now)."""
        if self.halo:
            self._peer_drain()
        self.stream.synchronize()
        if self.halo:
            if self.calc_stream2 is not self.stream:
                self.calc_stream2.synchronize()
            self.halo_stream.synchronize()
            if self.peer is not None:
                self.peer.check()       # a wait that gave up (neighbour gone): an error here, not wrong numbers later

    def initial_conditions(self):
        BoxSim.initial_conditions(self)
        if self.halo and self.xface is not None:
            self._reset_faces()
        if self.halo:
            self.sync()          # the first step starts on several streams
            if self.peer is not None:
                self.peer.barrier()      # ... and in several processes: nobody writes into a neighbour that is not ready

    def materialise_faces(self):
        """With x-face buffers the arrays are stale at the faces: write the received values into them (before anything
        reads the arrays on the host)."""
        if self.halo and self.xface is not None and self.iteration > 0:
            self.sync()
            last = self.iteration - 1
            pushed = (not self.aa) or (last & 1) == 1
            self.xface.materialise(self.gpu_dist[self.current_dist_index()], pushed, self.stream, parity=last & 1)
            self.sync()

    def _prime_pull(self):
        """A state written from the host at an odd in-place iteration: the next step pulls, and its edge lanes take what
        enters through a connected x face from the receive buffers alone (no pull out of the ghost column, slf_row.hip)
        -- so the ghost columns of the state just written go into those buffers."""
        if self.aa and (self.iteration & 1):
            self.xface.prime_pull(self.gpu_dist[0], self.stream, parity=1 - (self.iteration & 1))

    def get_dist(self, which=None):
        self.materialise_faces()
        return BoxSim.get_dist(self, which)

    def set_dist(self, host, which=None):
        """A state written from the host: whatever crossed the x faces before no longer counts."""
        BoxSim.set_dist(self, host, which)
        if self.halo and self.xface is not None:
            self._reset_faces()
            self._prime_pull()
        if self.halo:
            self.sync()
            if self.peer is not None:
                self.peer.barrier()

    # -- initial state ---------------------------------------------------------
    def init_synthetic(self, seed=1234):
        """rho = 1 + 1e-3 U[0,1), u = 0.05 (sin 2 pi y/Ly, sin 2 pi z/Lz, sin 2 pi x/Lx),
        coordinates measured in the global (all slabs) box."""
        nx, ny, nz = self.size
        g = [nx, ny, nz]
        g[self.axis] *= self.world
        o = [0, 0, 0]
        o[self.axis] = self.rank * self.size[self.axis]
        rng = np.random.RandomState(seed + self.rank)
        rho = (1.0 + 1e-3 * rng.rand(nz, ny, nx)).astype(self.dtype)
        x = np.arange(nx, dtype=np.float64) + o[0]
        y = np.arange(ny, dtype=np.float64) + o[1]
        z = np.arange(nz, dtype=np.float64) + o[2]
        vx = np.broadcast_to((0.05 * np.sin(2 * np.pi * y / g[1]))[None, :, None], (nz, ny, nx))
        vy = np.broadcast_to((0.05 * np.sin(2 * np.pi * z / g[2]))[:, None, None], (nz, ny, nx))
        vz = np.broadcast_to((0.05 * np.sin(2 * np.pi * x / g[0]))[None, None, :], (nz, ny, nx))
        self.set_fields(rho, [vx.astype(self.dtype), vy.astype(self.dtype), vz.astype(self.dtype)])
        self.initial_conditions()
        self.sync()
