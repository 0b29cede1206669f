"""Subdomain connectors: how halo buffers travel between neighbouring subdomains.

The reference moves halos device -> host -> zmq PAIR socket -> host -> device
(sailfish/connector.py:73-174, subdomain_runner.py:1064-1139).  Here the halo
stays on the device: one process per GPU, buffers exchanged with
torch.distributed point-to-point operations -- backend "nccl" (= RCCL over
xGMI) for GPU tensors, "gloo" for the CPU tests of the exchange logic.  No
collective is involved: a step needs one send + one receive per neighbour.
"""
import os


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK, WORLD_SIZE, MASTER_*)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1:
        return 0, 1
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    kw = {}
    if backend == 'nccl':
        kw['device_id'] = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world


class RingExchanger(object):
    """Nearest-neighbour exchange on a periodic 1-D ring of ranks (slab decomposition).

    exchange(send_up, send_down, recv_low, recv_high):
      send_up   -> rank+1, arrives in its recv_low
      send_down -> rank-1, arrives in its recv_high
    All four are torch tensors (CUDA for nccl, CPU for gloo).  Operations are
    batched (ncclGroupStart/End under the hood); with two ranks both messages go
    to the same peer and are matched by posting order (up first, then down).
    """

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.up = (rank + 1) % world
        self.down = (rank - 1) % world

    def exchange(self, send_up, send_down, recv_low, recv_high):
        import torch.distributed as dist
        if self.world == 1:
            recv_low.copy_(send_up)
            recv_high.copy_(send_down)
            return []
        ops = [dist.P2POp(dist.isend, send_up, self.up),
               dist.P2POp(dist.isend, send_down, self.down),
               dist.P2POp(dist.irecv, recv_low, self.down),
               dist.P2POp(dist.irecv, recv_high, self.up)]
        reqs = dist.batch_isend_irecv(ops)
        for r in reqs:
            r.wait()
        return reqs
