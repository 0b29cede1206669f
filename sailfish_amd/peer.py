"""Peer transport: the halo buffers of a subdomain process mapped into its neighbours' address space (C ABI slf_peer_*).

The reference moves a halo device -> pinned host -> zmq socket -> pinned host -> device (connector.py:73-174,
subdomain_runner.py:1064-1139) and left the hook for anything more direct unused (backend_cuda.py:122-126
ipc_handle / ipc_handle_wrap).  Inside one node every GPU reaches every other one's memory over xGMI, so here the
kernels that PRODUCE a halo -- the edge lanes of an x-split sweep, the pack kernels of the other faces -- store straight
into the receive buffers of the neighbouring PROCESS, and the exchange shrinks to ordering: one progress counter per
(sender, receiver, channel), advanced by a one-lane kernel behind the writes, awaited by a one-lane kernel on the
receiver's stream.  No copy, no RCCL kernels beside the sweep, nothing on the host; both entries are part of the step
plans.  RCCL (connector.DirectRccl) stays the transport where processes cannot map each other's memory.

Two processes that share ONE device map the same physical memory through the same calls, so the multi-process step runs
-- and can be measured -- on a single-GPU box (profiles/r06/ipc_probe.txt: eight processes, 13 us per counter hop, their
copy kernels share the device's bandwidth).

What travels over the side channel (torch.distributed, any backend; a ring of one needs none): 64-byte IPC handles, once.
"""
import ctypes
import os
import socket
import sys

from sailfish_amd import hipabi

CH_DIST, CH_MACRO, CH_ACK, CH_TEST = 0, 1, 2, 3      # channels: population halo, macro-field halo, self-test (2, 3)


class PeerUnavailable(RuntimeError):
    """The processes of this run cannot map each other's device memory (or the start-up check failed)."""


def _check(lib, status, what):
    from sailfish_amd.backend_hip import _check as chk
    chk(lib, status, what)


def _dist():
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if (dist.is_available() and dist.is_initialized()) else None


class PeerGroup(object):
    """The peer buffers of one simulation: allocated here, published once (collective), released together."""

    def __init__(self, transport):
        self.t = transport
        self._own = []           # device addresses allocated through this group
        self._opened = {}        # (rank, handle) -> where that buffer of another process is mapped here
        self._tables = None

    def alloc(self, nbytes):
        addr = self.t.alloc(nbytes)
        self._own.append(addr)
        return addr

    def publish(self, names):
        """names: {name: device address inside a buffer of alloc()}.  Collective over all ranks (every rank calls it once
        per simulation set-up, in the same order); afterwards lookup(rank, name) gives the address under which THIS
        process reaches the named buffer of `rank`."""
        mine = {}
        for name, addr in names.items():
            base, handle = self.t.containing(addr)
            mine[name] = (handle, int(addr) - base)
        self._tables = self.t.gather(mine)
        self._names = dict(names)

    def lookup(self, rank, name):
        if rank == self.t.rank:
            return self._names[name]
        handle, off = self._tables[rank][name]
        key = (rank, handle)
        if key not in self._opened:
            # mapped per group and closed with it: a handle names an ALLOCATION, and an allocation that is freed and made
            # again (the next simulation of the same process) may come back under the same handle -- a mapping kept
            # beyond its group would then point at memory nobody owns any more
            self._opened[key] = self.t.open(rank, handle)
        return self._opened[key] + off

    def release(self, collective=True):
        """Frees this group's buffers.  collective: every rank is doing the same right now, so wait until nobody can
        still be writing into them (the owner's streams are the caller's business)."""
        if collective and self.t.world > 1 and not sys.is_finalizing():
            try:
                self.t.barrier()
            except Exception:  # noqa: BLE001 -- a rank that has gone cannot hold the others back here
                pass
        for addr in self._opened.values():
            self.t.close_mapping(addr)
        self._opened = {}
        for addr in self._own:
            self.t.free(addr)
        self._own = []


class PeerTransport(object):
    """One per process (process_transport): the progress counters, the IPC handle cache, the entries a step program
    uses (stepqueue: peer_signal / peer_wait)."""

    def __init__(self, backend, rank, world, selftest=True):
        self.backend, self.rank, self.world = backend, int(rank), int(world)
        self.lib = backend._lib
        self._allocs = {}            # base address -> (bytes, handle)
        self._arrays = {}
        self.handle = None
        d = _dist()
        if self.world > 1 and d is None:
            raise PeerUnavailable('%d ranks but no torch.distributed process group to hand the IPC handles round' % world)
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.slf_peer_create(backend._ctx, self.world, self.rank, ctypes.byref(h)), 'slf_peer_create')
        self.handle = h
        verdict = self._connect_and_test(selftest)
        verdicts = self.gather(verdict)
        bad = ['rank %d: %s' % (r, v) for r, v in enumerate(verdicts) if v]
        if bad:
            self.close()
            raise PeerUnavailable('; '.join(bad))
        self.set_timeout(float(os.environ.get('SLF_PEER_TIMEOUT_S', '60')))

    # -- side channel ---------------------------------------------------------------------------------------------
    def gather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank."""
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        _dist().all_gather_object(out, obj)
        return out

    def barrier(self):
        if self.world > 1:
            _dist().barrier()

    # -- set-up -----------------------------------------------------------------------------------------------------
    def _connect_and_test(self, selftest):
        """'' or the reason this rank cannot use the transport.  Never raises: every rank must reach the collective
        verdict that follows."""
        try:
            buf = ctypes.create_string_buffer(hipabi.SLF_PEER_HANDLE_BYTES)
            _check(self.lib, self.lib.slf_peer_flags_handle(self.handle, buf), 'slf_peer_flags_handle')
            infos = self.gather((socket.gethostname(), buf.raw))
        except Exception as e:  # noqa: BLE001
            try:
                self.gather(None)
            except Exception:  # noqa: BLE001
                pass
            return 'handle exchange failed: %s' % str(e)[:150]
        reason = ''
        try:
            if any(i is None for i in infos):
                raise PeerUnavailable('a rank could not export its counters')
            hosts = set(i[0] for i in infos)
            if len(hosts) > 1:
                raise PeerUnavailable('ranks on different hosts (%s)' % ', '.join(sorted(hosts)))
            for r, (_, raw) in enumerate(infos):
                if r != self.rank:
                    _check(self.lib, self.lib.slf_peer_connect(self.handle, r, ctypes.create_string_buffer(raw, len(raw))),
                           'slf_peer_connect(rank %d)' % r)
        except Exception as e:  # noqa: BLE001
            reason = str(e)[:200]
        # everybody knows by now whether everybody is connected: the self-test needs all of its ring
        reasons = self.gather(reason)
        if any(reasons):
            return reason or 'another rank could not connect'
        if not selftest or os.environ.get('SLF_PEER_SELFTEST', '1') == '0':
            return ''
        try:
            return self._selftest()
        except Exception as e:  # noqa: BLE001
            return 'self-test failed: %s' % str(e)[:200]

    def _selftest(self, nwords=1 << 16, rounds=3):
        """Plain device memory of the ring neighbour written through its mapping (dense stores and one word per workgroup,
        the x-face pattern) and released by a signal is what the neighbour reads after its wait -- on the mappings and
        streams of this very run, with the reader holding the previous contents in its caches."""
        self.set_timeout(float(os.environ.get('SLF_PEER_SELFTEST_TIMEOUT_S', '60')))
        grp = PeerGroup(self)
        mine = grp.alloc(nwords * 4)
        self.backend.memset_buf(mine, 0, nwords * 4)
        self.backend.sync()
        grp.publish({'selftest': mine})
        up, down = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        target = grp.lookup(up, 'selftest')
        s = self.backend.make_stream()
        bad_total = 0
        out = ctypes.c_uint32()
        # the first launch of a process loads the library's code objects (seconds when eight processes start on one device
        # at once): do that outside the timed waits, then start the rounds together
        _check(self.lib, self.lib.slf_peer_selftest_check(self.handle, ctypes.c_void_p(mine), 64, 0, s.handle, ctypes.byref(out)),
               'slf_peer_selftest_check')
        self.signal([self.rank], CH_ACK, s)
        self.wait([self.rank], CH_ACK, s)
        s.synchronize()
        self.barrier()
        for sparse in (0, 1):
            for rnd in range(rounds):
                pat = lambda r: (0x9e3779b9 * (1 + rnd + 100 * sparse) + 7919 * r) & 0xFFFFFFFF   # noqa: E731
                _check(self.lib, self.lib.slf_peer_selftest_fill(self.handle, ctypes.c_void_p(target), nwords, pat(self.rank),
                                                                 sparse, s.handle), 'slf_peer_selftest_fill')
                self.signal([up], CH_TEST, s)
                self.wait([down], CH_TEST, s)
                _check(self.lib, self.lib.slf_peer_selftest_check(self.handle, ctypes.c_void_p(mine), nwords, pat(down),
                                                                  s.handle, ctypes.byref(out)), 'slf_peer_selftest_check')
                bad_total += int(out.value)
                # the neighbour below may overwrite my buffer again once I have read it, and so may I the one above
                self.signal([down], CH_ACK, s)
                self.wait([up], CH_ACK, s)
        s.synchronize()
        st = self.status()
        grp.release(collective=False)
        if st['timed_out']:
            return 'self-test: no signal from rank %d within the time-out (expected %d, seen %d)' % (
                st['rank'], st['expected'], st['seen'])
        if bad_total:
            return 'self-test: %d words written through the mapping arrived wrong' % bad_total
        return ''

    # -- memory -------------------------------------------------------------------------------------------------------
    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        h = ctypes.create_string_buffer(hipabi.SLF_PEER_HANDLE_BYTES)
        _check(self.lib, self.lib.slf_peer_alloc(self.handle, max(1, int(nbytes)), ctypes.byref(p), h), 'slf_peer_alloc')
        self._allocs[p.value] = (max(1, int(nbytes)), h.raw)
        return p.value

    def free(self, addr):
        if self._allocs.pop(addr, None) is not None and self.handle is not None:
            self.lib.slf_peer_free(self.handle, ctypes.c_void_p(addr))

    def containing(self, addr):
        """(base address, handle) of the alloc() buffer `addr` lies in."""
        for base, (n, handle) in self._allocs.items():
            if base <= addr < base + n:
                return base, handle
        raise KeyError('address %#x is not inside a peer buffer of this process' % addr)

    def open(self, rank, handle):
        """Maps a buffer of another process (whole allocation); the caller closes it (PeerGroup does)."""
        p = ctypes.c_void_p()
        _check(self.lib, self.lib.slf_peer_open(self.handle, ctypes.create_string_buffer(handle, len(handle)), ctypes.byref(p)),
               'slf_peer_open(rank %d)' % rank)
        return p.value

    def close_mapping(self, addr):
        if self.handle is not None:
            self.lib.slf_peer_close(self.handle, ctypes.c_void_p(addr))

    def group(self):
        return PeerGroup(self)

    # -- ordering -----------------------------------------------------------------------------------------------------
    def ranks_array(self, ranks):
        key = tuple(int(r) for r in ranks)
        arr = self._arrays.get(key)
        if arr is None:
            arr = self._arrays[key] = (ctypes.c_int32 * max(1, len(key)))(*key)
        return arr, len(key)

    def signal(self, ranks, channel, stream):
        arr, n = self.ranks_array(ranks)
        _check(self.lib, self.lib.slf_peer_signal(self.handle, arr, n, int(channel), stream.handle), 'slf_peer_signal')

    def wait(self, ranks, channel, stream, count=1):
        """`stream` continues once `count` further signals of every rank of `ranks` have arrived."""
        arr, n = self.ranks_array(ranks)
        _check(self.lib, self.lib.slf_peer_wait(self.handle, arr, n, int(channel), int(count), stream.handle), 'slf_peer_wait')

    def set_timeout(self, seconds):
        _check(self.lib, self.lib.slf_peer_set_timeout(self.handle, float(seconds)), 'slf_peer_set_timeout')

    def status(self):
        out = (ctypes.c_int64 * 8)()
        _check(self.lib, self.lib.slf_peer_status(self.handle, ctypes.byref(out)), 'slf_peer_status')
        return {'timed_out': bool(out[0]), 'rank': int(out[1]), 'channel': int(out[2]), 'expected': int(out[3]),
                'seen': int(out[4]), 'timeouts': int(out[5])}

    def check(self):
        """Raises when a wait of this process has given up (call where the host synchronises anyway)."""
        st = self.status()
        if st['timed_out']:
            from sailfish_amd.backend_hip import HIPFatalError
            raise HIPFatalError('peer transport: rank %d waited for signal %d of rank %d on channel %d and saw %d when it '
                                'gave up -- the neighbour is gone or stuck' % (self.rank, st['expected'], st['rank'],
                                                                                st['channel'], st['seen']))

    def progress(self, rank, channel=CH_DIST):
        """{'sent', 'awaited', 'arrived'} for one neighbour: what the watchdog prints about a run that stopped."""
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _check(self.lib, self.lib.slf_peer_progress(self.handle, int(rank), int(channel), ctypes.byref(a), ctypes.byref(b),
                                                    ctypes.byref(c)), 'slf_peer_progress')
        return {'sent': a.value, 'awaited': b.value, 'arrived': c.value}

    def snapshot(self, ranks=None):
        """Host-side facts only (no device call: usable from a watchdog thread while the device is stuck): the last
        time-out, and per neighbour and channel how many signals this process has enqueued / is waiting for."""
        out = {'status': self.status(), 'pairs': []}
        for r in (ranks if ranks is not None else range(self.world)):
            for ch in range(hipabi.SLF_PEER_CHANNELS):
                a, b = ctypes.c_uint64(), ctypes.c_uint64()
                if self.lib.slf_peer_progress(self.handle, int(r), ch, ctypes.byref(a), ctypes.byref(b), None) == 0 and \
                        (a.value or b.value):
                    out['pairs'].append({'rank': int(r), 'channel': ch, 'signals_enqueued': a.value, 'waits_enqueued': b.value})
        return out

    def close(self):
        if self.handle is not None:
            for addr in list(self._allocs):
                self.free(addr)
            self.lib.slf_peer_destroy(self.handle)
            self.handle = None


_process = {}
unavailable_reason = None


def requested():
    """SLF_HALO_TRANSPORT: auto (default: peer where the processes can map each other's memory, else RCCL, else
    torch.distributed), peer, rccl, torch."""
    return os.environ.get('SLF_HALO_TRANSPORT', 'auto')


def process_transport(backend, rank, world):
    """The PeerTransport of this process, or None when the run's processes cannot use one (reason in
    `unavailable_reason`; SLF_HALO_TRANSPORT=peer makes that an error).  Creating it is a collective over all ranks:
    once per process, and every rank asks at the same point of its set-up."""
    global unavailable_reason
    if requested() not in ('auto', 'peer') or not hasattr(backend, '_ctx'):
        return None
    key = (int(backend.gpu_id), int(rank), int(world))
    if key not in _process:
        try:
            _process[key] = PeerTransport(backend, rank, world)
        except PeerUnavailable as e:
            unavailable_reason = str(e)
            _process[key] = None
            if requested() == 'peer':
                raise
            sys.stderr.write('sailfish_amd: peer transport unavailable (%s); halos travel over RCCL / torch.distributed\n' % e)
    return _process[key]
