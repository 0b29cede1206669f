"""Subdomain connectors: how halo buffers travel between neighbouring subdomains.

The reference moves halos device -> host -> zmq PAIR socket -> host -> device
(sailfish/connector.py:73-174, subdomain_runner.py:1064-1139).  Here the halo
stays on the device: one process per GPU, buffers exchanged with
torch.distributed point-to-point operations -- backend "nccl" (= RCCL over
xGMI) for GPU tensors, "gloo" for the CPU tests of the exchange logic.  No
collective is involved: a step needs one send + one receive per neighbour.
"""
import os


def init_distributed(backend=None, force=False):
    """Initialise torch.distributed from the torchrun environment (RANK, WORLD_SIZE, MASTER_*).  A single rank
    needs no process group unless `force` (bench.py --force_distributed: the RCCL path with one rank)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1 and not force:
        return 0, 1
    if backend is None:
        # SLF_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses two ranks per device); device tensors are
        # then staged through the host by the exchangers -- a functional stand-in for tests, not a fast path
        backend = os.environ.get('SLF_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    kw = {}
    if backend == 'nccl':
        # one process per GPU: make that GPU torch's current device too (streams handed to torch and RCCL's
        # own stream must live on it)
        local = int(os.environ.get('SLF_FORCE_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        torch.cuda.set_device(local)
        kw['device_id'] = torch.device('cuda', local)
        # RCCL's own stream ahead of the bulk sweep as well (SLF_HALO_PRIORITY=0: the A/B switch)
        if os.environ.get('SLF_HALO_PRIORITY', '1') != '0' and hasattr(dist, 'ProcessGroupNCCL'):
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            kw['pg_options'] = opts
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world


class NoQuiesce(object):
    def quiesce(self, runner):
        """Nobody else writes into this runner's halo buffers behind its back: nothing to wait for."""

    def release(self, runner):
        pass


class RingExchanger(object):
    """Nearest-neighbour exchange on a periodic 1-D ring of ranks (slab decomposition).

    exchange(send_up, send_down, recv_low, recv_high):
      send_up   -> rank+1, arrives in its recv_low
      send_down -> rank-1, arrives in its recv_high
    All four are torch tensors (CUDA for nccl, CPU for gloo).  Operations are
    batched (ncclGroupStart/End under the hood); with two ranks (or one: send / recv
    to self) both messages go to the same peer and are matched by posting order (up
    first, then down).
    """

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.up = (rank + 1) % world
        self.down = (rank - 1) % world

    def plain_copy(self):
        """A ring of one without a process group: the exchange is two device copies (which a step plan can hold)."""
        import torch.distributed as dist
        return self.world == 1 and not (dist.is_available() and dist.is_initialized())

    def exchange_range(self, bufs, start, count):
        """The same exchange for the elements [start, start + count) of the four buffers (x-face buffers: a range of
        z-planes that the sweep has completed, sailfish_amd/xface.py)."""
        return self.exchange(*[t[start:start + count] for t in bufs])

    def exchange(self, send_up, send_down, recv_low, recv_high):
        import torch
        import torch.distributed as dist
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            recv_low.copy_(send_up)       # a ring of one without a process group: the slab is its own neighbour
            recv_high.copy_(send_down)
            return []
        staged = send_up.is_cuda and dist.get_backend() == 'gloo'
        if staged:      # gloo moves host memory only (see init_distributed)
            dev = (recv_low, recv_high)
            send_up, send_down = send_up.cpu(), send_down.cpu()
            recv_low, recv_high = torch.empty_like(send_up), torch.empty_like(send_down)
        ops = [dist.P2POp(dist.isend, send_up, self.up),
               dist.P2POp(dist.isend, send_down, self.down),
               dist.P2POp(dist.irecv, recv_low, self.down),
               dist.P2POp(dist.irecv, recv_high, self.up)]
        reqs = dist.batch_isend_irecv(ops)
        for r in reqs:
            r.wait()
        if staged:
            dev[0].copy_(recv_low)
            dev[1].copy_(recv_high)
        return reqs


class DirectRccl(object):
    """One RCCL communicator per process through the C ABI (slf_comm_*, include/sailfish_hip.h): send / receive are two
    ctypes calls each, where torch.distributed builds P2POp lists, a coalescing manager and work objects per group --
    hundreds of microseconds of host time per exchange, which is most of a 0.9 ms step once the halo goes out in
    several batches.  The 128-byte communicator id is created by rank 0 and handed round through the torch.distributed
    process group that the launcher set up anyway (the side channel the C ABI asks for)."""

    def __init__(self, backend, rank, world):
        import ctypes
        import torch.distributed as dist
        from sailfish_amd.backend_hip import _check
        self._ctypes, self._check, self.lib = ctypes, _check, backend._lib
        self._backend = backend          # the communicator refers to the backend's context: keep it alive
        uid = ctypes.create_string_buffer(128)
        if rank == 0:
            _check(self.lib, self.lib.slf_comm_unique_id(uid), 'slf_comm_unique_id')
        if world > 1:
            box = [uid.raw]
            dist.broadcast_object_list(box, src=0)
            uid = ctypes.create_string_buffer(box[0], 128)
        self.comm = ctypes.c_void_p()
        _check(self.lib, self.lib.slf_comm_init(backend._ctx, int(world), int(rank), uid, ctypes.byref(self.comm)),
               'slf_comm_init')

    def prepare(self, ops):
        """ops: [('send' | 'recv', peer, device address, elements, element bytes)] -> a batch that run() posts in this
        order inside ONE RCCL group (C ABI slf_comm_exchange: one call per batch; the halo batches of a simulation are
        the same every step, so they are built once)."""
        from sailfish_amd import hipabi
        arr = (hipabi.SlfCommOp * max(1, len(ops)))()
        for i, (what, peer, addr, n, isz) in enumerate(ops):
            arr[i].kind = 0 if what == 'send' else 1
            arr[i].peer, arr[i].dptr, arr[i].count, arr[i].elem_bytes = int(peer), int(addr), int(n), int(isz)
        return (arr, len(ops))

    def run(self, batch, stream):
        arr, n = batch
        self._check(self.lib, self.lib.slf_comm_exchange(self.comm, arr, n, stream.handle), 'slf_comm_exchange')

    def group(self, ops, stream):
        self.run(self.prepare(ops), stream)

    def count(self):
        """(ranks, my rank) as RCCL reports them for this communicator (ncclCommCount / ncclCommUserRank)."""
        n, r = self._ctypes.c_int(), self._ctypes.c_int()
        self._check(self.lib, self.lib.slf_comm_count(self.comm, self._ctypes.byref(n), self._ctypes.byref(r)), 'slf_comm_count')
        return int(n.value), int(r.value)

    def close(self):
        if self.comm:
            self.lib.slf_comm_destroy(self.comm)
            self.comm = None


_process_rccl = {}


def process_rccl(backend, rank, world):
    """The one DirectRccl communicator of this process (creating one is a collective over all ranks: once, not per
    simulation)."""
    key = (int(backend.gpu_id), int(rank), int(world))
    if key not in _process_rccl:
        _process_rccl[key] = DirectRccl(backend, rank, world)
    return _process_rccl[key]


class RcclRingExchanger(RingExchanger):
    """RingExchanger whose transfers go straight to RCCL (DirectRccl) on a stream of the backend."""
    direct = True

    def __init__(self, rank, world, backend):
        RingExchanger.__init__(self, rank, world)
        self.rccl = process_rccl(backend, rank, world)
        self._batches = {}

    def exchange_ranges(self, bufs, ranges, stream):
        self.rccl.run(self.batch(bufs, ranges), stream)

    def batch(self, bufs, ranges):
        """bufs = (send_up, send_down, recv_low, recv_high) tensors; ranges = [(first element, count)]: one group.
        Posting order = the matching order between a pair of ranks: up first, then down (a ring of two has both
        messages going to the same peer).  The batch is built once per (buffers, ranges)."""
        key = (tuple(t.data_ptr() for t in bufs), tuple(ranges))
        batch = self._batches.get(key)
        if batch is None:
            s_up, s_down, r_low, r_high = bufs
            isz = s_up.element_size()
            ops = []
            for start, count in ranges:
                off = start * isz
                ops += [('send', self.up, s_up.data_ptr() + off, count, isz), ('send', self.down, s_down.data_ptr() + off, count, isz),
                        ('recv', self.down, r_low.data_ptr() + off, count, isz), ('recv', self.up, r_high.data_ptr() + off, count, isz)]
            batch = self._batches[key] = self.rccl.prepare(ops)
        return batch


def make_ring_exchanger(rank, world, backend):
    """The transport of the slab ring: RCCL through the C ABI where the process group is RCCL ("nccl"), otherwise
    torch.distributed (gloo stages through the host: several ranks on one GPU in the tests), a plain copy for a ring
    of one without a process group.  SLF_HALO_TRANSPORT=torch forces torch.distributed."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl' and \
            os.environ.get('SLF_HALO_TRANSPORT', 'rccl') != 'torch':
        try:
            return RcclRingExchanger(rank, world, backend)
        except Exception as e:  # noqa: BLE001 -- no communicator of our own: torch.distributed carries the halo instead
            import sys
            sys.stderr.write('sailfish_amd: RCCL through the C ABI unavailable (%s); halo through torch.distributed\n' % e)
    return RingExchanger(rank, world)


class LocalConnector(NoQuiesce):
    """All subdomains live in this process (one or several GPUs): halos move with device-to-device
    copies driven by the group driver (controller.LocalGroup); the reference's counterpart is
    MPSubdomainConnector (connector.py:120-174), which stages through shared host memory."""

    def alloc_buffer(self, runner, nelems, dtype):
        import numpy as np
        return runner.backend.alloc_buf(size=max(1, nelems) * np.dtype(dtype).itemsize)

    def exchange(self, runner):
        raise RuntimeError('LocalConnector exchanges are driven by controller.LocalGroup')

    def enqueue_exchange(self, q, runner, kind='dist'):
        if runner.halo_messages(kind):
            raise RuntimeError('LocalConnector exchanges are driven by controller.LocalGroup')

    def enqueue_pieces(self, q, runner, pieces):
        if pieces:
            raise RuntimeError('LocalConnector exchanges are driven by controller.LocalGroup')


class TorchDistConnector(NoQuiesce):
    """Neighbours live in other processes (one process per GPU): point-to-point send / receive of the
    packed halo buffers with torch.distributed (RCCL over xGMI for CUDA tensors, gloo for CPU tensors in
    the tests).  Replaces ZMQSubdomainConnector (reference connector.py:73-117)."""

    def __init__(self, id_to_rank, device=None):
        self.id_to_rank = dict(id_to_rank)
        self.device = device
        self._tensors = {}
        self._stream = None

    def alloc_buffer(self, runner, nelems, dtype):
        import numpy as np
        import torch
        tdtype = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
        dev = self.device
        if dev is None:     # where the backend's kernels can reach memory: the GPU of the runner (CPU test backends say so)
            dev = torch.device(getattr(runner.backend, 'tensor_device', None) or ('cuda:%d' % runner.backend.gpu_id))
        t = torch.empty(max(1, nelems), dtype=tdtype, device=dev)
        self._tensors[t.data_ptr()] = t
        return t.data_ptr()

    def tensor(self, ptr):
        return self._tensors[ptr]

    @staticmethod
    def exchange_tensors(sends, recvs):
        """sends / recvs: lists of (tensor, peer rank), each ordered by neighbour subdomain id.  One
        batched group of point-to-point operations (no collective); returns after the operations have
        been *enqueued* for CUDA tensors (stream-ordered) and completed for CPU tensors."""
        import torch
        import torch.distributed as dist
        staged = []
        if dist.get_backend() == 'gloo' and any(t.is_cuda for t, _ in sends + recvs):
            # several ranks on one GPU (init_distributed: SLF_DIST_BACKEND=gloo): gloo moves host memory only
            sends = [(t.cpu(), peer) for t, peer in sends]
            staged = [(t, torch.empty(t.shape, dtype=t.dtype)) for t, _ in recvs]
            recvs = [(h, peer) for (_, h), (_, peer) in zip(staged, recvs)]
        ops = [dist.P2POp(dist.isend, t, peer) for t, peer in sends]
        ops += [dist.P2POp(dist.irecv, t, peer) for t, peer in recvs]
        if not ops:
            return
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        for dev, host in staged:
            dev.copy_(host)

    mid_step = True      # exchange_pieces() may be called while the step is still being enqueued

    def view(self, addr, n):
        """The n elements at device address `addr`, somewhere inside a buffer handed out by alloc_buffer()."""
        t = self._tensors.get(addr)
        if t is not None:
            return t[:n]
        for base, t in self._tensors.items():
            off = addr - base
            if 0 <= off < t.numel() * t.element_size():
                i = off // t.element_size()
                return t[i:i + n]
        raise KeyError('address %#x is not inside a halo buffer of this connector' % addr)

    _rccl = None

    @property
    def _batches(self):
        return self.__dict__.setdefault('_batch_cache', {})

    def direct(self, runner):
        """DirectRccl for this process (None where the process group is not RCCL, or SLF_HALO_TRANSPORT=torch)."""
        import torch.distributed as dist
        if self._rccl is None:
            self._rccl = False
            if dist.get_backend() == 'nccl' and os.environ.get('SLF_HALO_TRANSPORT', 'rccl') != 'torch' and \
                    hasattr(runner.backend, '_ctx'):
                self._rccl = process_rccl(runner.backend, dist.get_rank(), dist.get_world_size())
        return self._rccl or None

    def enqueue_exchange(self, q, runner, kind='dist'):
        """The exchange of `kind` that is due now as an entry of the step program `q` (stepqueue.py): one RCCL group
        through the C ABI where the process group is RCCL -- something a step plan can hold -- otherwise a call back
        into exchange() (torch.distributed; entry-by-entry queues only)."""
        msgs = runner.halo_messages(kind)
        if not msgs:
            return
        rccl = self.direct(runner)
        if rccl is None:
            q.call(lambda: self.exchange(runner, kind))
            return
        key = (kind, tuple(msgs))
        batch = self._batches.get(key)
        if batch is None:
            isz = runner.float().itemsize
            ops = [('send', self.id_to_rank[nid], sb, ns, isz) for nid, sb, ns, _, _ in msgs if ns]
            ops += [('recv', self.id_to_rank[nid], rb, nr, isz) for nid, _, _, rb, nr in msgs if nr]
            batch = self._batches[key] = rccl.prepare(ops)
        q.exchange(rccl, batch, runner._data_stream)

    def enqueue_pieces(self, q, runner, pieces):
        """exchange_pieces() as an entry of the step program `q`."""
        if not pieces:
            return
        rccl = self.direct(runner)
        if rccl is None:
            q.call(lambda: self.exchange_pieces(runner, pieces))
            return
        key = tuple(pieces)
        batch = self._batches.get(key)
        if batch is None:
            isz = runner.float().itemsize
            ops = [('send', self.id_to_rank[nid], s, n, isz) for nid, s, _, n in pieces]
            ops += [('recv', self.id_to_rank[nid], r, n, isz) for nid, _, r, n in pieces]
            batch = self._batches[key] = rccl.prepare(ops)
        q.exchange(rccl, batch, runner._data_stream)

    def exchange_pieces(self, runner, pieces):
        """pieces: [(neighbour id, send address, receive address, elements)] -- ranges of the x-face buffers that a
        z-chunk of the sweep has completed (subdomain_runner._run_sweep_xface); one batched group, on the data stream."""
        import torch
        if not pieces:
            return
        rccl = self.direct(runner)
        if rccl is not None:
            key = tuple(pieces)
            batch = self._batches.get(key)
            if batch is None:          # the same few batches every step: built once
                isz = runner.float().itemsize
                ops = [('send', self.id_to_rank[nid], s, n, isz) for nid, s, _, n in pieces]
                ops += [('recv', self.id_to_rank[nid], r, n, isz) for nid, _, r, n in pieces]
                batch = self._batches[key] = rccl.prepare(ops)
            rccl.run(batch, runner._data_stream)
            return
        sends = [(self.view(s, n), self.id_to_rank[nid]) for nid, s, _, n in pieces]
        recvs = [(self.view(r, n), self.id_to_rank[nid]) for nid, _, r, n in pieces]
        if not pieces:
            return
        if not sends[0][0].is_cuda:
            self.exchange_tensors(sends, recvs)
            return
        if self._stream is None:
            self._stream = torch.cuda.ExternalStream(runner._data_stream.native,
                                                     device=torch.device('cuda', runner.backend.gpu_id))
        with torch.cuda.stream(self._stream):
            self.exchange_tensors(sends, recvs)

    def exchange(self, runner, kind='dist'):
        import torch
        msgs = runner.halo_messages(kind)
        rccl = self.direct(runner) if msgs else None
        if rccl is not None:
            isz = runner.float().itemsize
            ops = [('send', self.id_to_rank[nid], sb, ns, isz) for nid, sb, ns, _, _ in msgs if ns]
            ops += [('recv', self.id_to_rank[nid], rb, nr, isz) for nid, _, _, rb, nr in msgs if nr]
            rccl.group(ops, runner._data_stream)
            return
        sends, recvs = [], []
        for nid, send_buf, n_send, recv_buf, n_recv in msgs:
            peer = self.id_to_rank[nid]
            if n_send:
                sends.append((self.tensor(send_buf)[:n_send], peer))
            if n_recv:
                recvs.append((self.tensor(recv_buf)[:n_recv], peer))
        if not sends and not recvs:
            return
        if not (sends + recvs)[0][0].is_cuda:      # CPU tensors (gloo): no stream to order against
            self.exchange_tensors(sends, recvs)
            return
        if self._stream is None:
            self._stream = torch.cuda.ExternalStream(runner._data_stream.native,
                                                     device=torch.device('cuda', runner.backend.gpu_id))
        with torch.cuda.stream(self._stream):
            self.exchange_tensors(sends, recvs)


class PeerConnector(TorchDistConnector):
    """Neighbours live in other processes of this node and every process can map the others' device memory
    (sailfish_amd/peer.py): a halo is never sent.  Each runner allocates its RECEIVE buffers -- two sets that alternate
    by step parity -- and publishes them; the buffers its pack kernels (or the edge lanes of its x-split sweep) write
    into are the neighbours' receive buffers, mapped here.  What remains of the exchange is a pair of plan entries:
    signal "my writes of this step are in your memory", wait for the neighbours' signal.

    Why two sets are enough (general faces; the x-face buffers add xface.ChunkPlan.peer_need): set p is written by my
    pack of step it (p = it & 1) and read by the neighbour's unpack of step it; it is written again by my pack of step
    it + 2, which follows -- on my data stream -- my wait for the neighbour's signal of step it + 1, and the neighbour
    enqueued that signal behind its unpack of step it on ITS data stream."""
    zero_copy = True
    mid_step = True

    def __init__(self, id_to_rank, transport):
        TorchDistConnector.__init__(self, id_to_rank)
        self.peer = transport
        self._groups = {}

    def _group(self, runner):
        g = self._groups.get(id(runner))
        if g is None:
            g = self._groups[id(runner)] = {'group': self.peer.group(), 'names': {}, 'published': False}
        return g

    def alloc_recv(self, runner, kind, nid, parity, nelems, dtype):
        """My receive buffer for what subdomain `nid` sends me (`kind`: 'dist' | 'macro') in the steps of `parity`."""
        import numpy as np
        g = self._group(runner)
        nbytes = max(1, int(nelems)) * np.dtype(dtype).itemsize
        addr = g['group'].alloc(nbytes)
        g['names'][(kind, int(runner._spec.id), int(nid), int(parity))] = addr
        return addr

    def resolve(self, runner):
        """Collective (every rank once per call site of its set-up, in the same order): afterwards send_addr() works."""
        g = self._group(runner)
        g['group'].publish(g['names'])

    def send_addr(self, runner, kind, nid, parity):
        """Where this process reaches the buffer in which subdomain `nid` receives what `runner` sends it."""
        g = self._group(runner)
        return g['group'].lookup(self.id_to_rank[nid], (kind, int(nid), int(runner._spec.id), int(parity)))

    def _ranks(self, nids):
        return sorted(set(self.id_to_rank[n] for n in nids))

    def enqueue_exchange(self, q, runner, kind='dist'):
        from sailfish_amd import peer as peer_mod
        nids = [m[0] for m in runner.halo_messages(kind) if m[2] or m[4]]
        if not nids:
            return
        ch = peer_mod.CH_DIST if kind == 'dist' else peer_mod.CH_MACRO
        q.peer_signal(self.peer, self._ranks(nids), ch, runner._data_stream)
        q.peer_wait(self.peer, self._ranks(nids), ch, runner._data_stream)

    def enqueue_pieces(self, q, runner, pieces):
        """x-face buffers: nothing to move, the planes are in the neighbours' memory when the chunk is done."""
        from sailfish_amd import peer as peer_mod
        ranks = self._ranks(nid for nid, _ in runner._xface_routes)
        q.peer_signal(self.peer, ranks, peer_mod.CH_DIST, runner._data_stream)
        q.peer_wait(self.peer, ranks, peer_mod.CH_DIST, runner._data_stream)

    def exchange(self, runner, kind='dist'):
        from sailfish_amd.stepqueue import DirectQueue
        self.enqueue_exchange(DirectQueue(runner.backend), runner, kind)

    def exchange_pieces(self, runner, pieces):
        from sailfish_amd.stepqueue import DirectQueue
        self.enqueue_pieces(DirectQueue(runner.backend), runner, pieces)

    def quiesce(self, runner):
        """The neighbours write into this runner's buffers themselves: before the host touches them (initial state, a
        restored checkpoint) every process has stopped stepping, and nobody starts again before all are done."""
        runner.backend.sync_stream(*runner._all_streams())
        self.peer.check()
        self.peer.barrier()

    def release(self, runner):
        g = self._groups.pop(id(runner), None)
        if g is not None:
            # not collective (runners are let go of wherever their owner pleases): a neighbour that still has a buffer
            # mapped keeps the memory behind it alive until it closes the mapping
            g['group'].release(collective=False)


def make_connector(id_to_rank, backend, rank, world):
    """The connector of a runner that owns its process: peer mappings where the processes of the run can map each other's
    memory (collective: every rank asks here), else torch.distributed / RCCL."""
    from sailfish_amd import peer as peer_mod
    pt = peer_mod.process_transport(backend, rank, world)
    if pt is not None:
        return PeerConnector(id_to_rank, pt)
    return TorchDistConnector(id_to_rank)
