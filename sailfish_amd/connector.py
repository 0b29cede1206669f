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


class LocalConnector(object):
    """All subdomains live in this process (one or several GPUs): halos move with device-to-device
    copies driven by the group driver (controller.LocalGroup); the reference's counterpart is
    MPSubdomainConnector (connector.py:120-174), which stages through shared host memory."""

    def alloc_buffer(self, runner, nelems, dtype):
        import numpy as np
        return runner.backend.alloc_buf(size=max(1, nelems) * np.dtype(dtype).itemsize)

    def exchange(self, runner):
        raise RuntimeError('LocalConnector exchanges are driven by controller.LocalGroup')


class TorchDistConnector(object):
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

    def exchange(self, runner, kind='dist'):
        import torch
        sends, recvs = [], []
        for nid, send_buf, n_send, recv_buf, n_recv in runner.halo_messages(kind):
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
