"""One time step as a program: the entries of a step (kernel launches with their regions and streams, event records,
stream waits, halo exchanges, buffer fills and copies) are issued against ONE interface, so that the step is written
once (`SlabSim._program`, `SubdomainRunner._program`) and is either

  * performed entry by entry through the backend calls (`DirectQueue`: transports that need Python in the middle of a
    step -- torch.distributed / gloo in the tests -- and steps that record timing events), or
  * recorded once into a C-ABI step plan (`backend_hip.HIPPlan`, include/sailfish_hip.h "step plans") and replayed
    with one call per step.

The reference enqueues every step from Python (subdomain_runner.py:960-974, 1028-1058) at ~15 ms per step; a
halo-connected MI355X subdomain steps in under a millisecond with ~20 runtime calls per step.
"""


class NotPlannable(RuntimeError):
    """An entry that only Python can perform (a torch.distributed exchange) was put into a step plan."""


class DirectQueue(object):
    """Performs every entry at once through the backend interface (any backend: the CPU test backend too)."""
    planned = False

    def __init__(self, backend):
        self.backend = backend

    def _of(self, stream):
        """The backend a stream belongs to (a same-process group steps runners with a backend object each)."""
        return getattr(stream, '_backend', None) or self.backend

    def launch(self, kernel, region, stream):
        self._of(stream).run_kernel(kernel, region, stream)

    def record(self, event, stream):
        event.record(stream)

    def wait(self, stream, event):
        stream.wait_for_event(event)

    def exchange(self, rccl, batch, stream):
        """batch: DirectRccl.prepare(ops) -- one RCCL group"""
        rccl.run(batch, stream)

    def peer_signal(self, peer, ranks, channel, stream):
        """peer transport (sailfish_amd/peer.py): what this stream has written into the buffers of `ranks` is complete"""
        peer.signal(ranks, channel, stream)

    def peer_wait(self, peer, ranks, channel, stream, count=1):
        """the stream continues once `count` further signals of every rank of `ranks` have arrived"""
        peer.wait(ranks, channel, stream, count)

    def memset(self, addr, value, nbytes, stream):
        self._of(stream).memset_buf(addr, value, nbytes, stream)

    def copy(self, dst, src, nbytes, stream):
        self._of(stream).copy_buf_async(dst, src, nbytes, stream)

    def xface(self, module, send_low, send_high, recv_low, recv_high):
        self.backend.set_xface_buffers(module, send_low, send_high, recv_low, recv_high)

    def xface_planes(self, module, which, send_low, send_high, recv_low, recv_high):
        self.backend.set_xface_planes(module, which, send_low, send_high, recv_low, recv_high)

    def call(self, fn):
        """Arbitrary host code between the entries (what a plan cannot hold)."""
        fn()
