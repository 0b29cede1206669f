"""Placement of the distribution arrays in HBM (MI355X-specific; no counterpart in the reference).

What was measured (profiles/r02/README.md): the sweep keeps 2 x Q = 38 streams open in HBM at once.  On MI355X the
rate it sustains depends on WHERE in physical memory the array it writes lies: 5.3 TB/s when the whole array sits
inside one coarse region (tens of GiB) of the physical address space, 6.1-6.4 TB/s when it is spread over several,
best with the traffic split evenly.  Single-stream kernels (copy, fill, read) do not care, the stride between the
direction arrays does not matter, clocks / power / TLB counters are identical in both cases -- only the number of
outstanding memory requests differs.  A plain hipMalloc puts a 10 GB array into one region about 40 % of the time.

What is done about it: a large distribution array is a *placed* buffer -- one reserved virtual range (so the layout
the kernels and the host see is the reference's `q * dist_stride + node`), backed by PARTS separately created
physical chunks.  While the chunks are created, spacer allocations are put between them and released afterwards, so
the chunks end up spread over SPAN bytes of physical memory, whatever the regions are; nothing is wasted once the
spacers are gone.  Several buffers placed together (the two copies of the AB pattern) share the span.  Skipped with
SLF_PLACEMENT=0 and for arrays below MIN_BYTES: measured on D2Q9 / D3Q19 boxes of 150 MB .. 2.4 GB per array
(profiles/r02/placement_threshold.log), placement gains 20 % at 600 MB (D2Q9 4096^2), is neutral at 570 MB (D3Q19 192^3)
and costs 8-20 % for arrays of 340 MB and less, which live partly in the 256 MB memory-side cache.
"""
import ctypes
import os

MIN_BYTES = int(os.environ.get('SLF_PLACEMENT_MIN_MIB', 512)) << 20
PARTS = 16
SPAN = 72 << 30


def enabled():
    return os.environ.get('SLF_PLACEMENT', '1') not in ('0', 'off', 'no')


def sharers():
    """How many processes place arrays on THIS device at the same time: 1 in production (one process per GPU); the
    world size when SLF_FORCE_DEVICE puts every rank of a functional run on one GPU (tests/test_gpu_two_ranks.py,
    config 4's eight subdomains on a 1-GPU box), or whatever SLF_DEVICE_SHARERS says.  Each then takes its share of the
    span and of the free memory, so that the spacers of all of them fit side by side."""
    n = os.environ.get('SLF_DEVICE_SHARERS')
    if n is None and os.environ.get('SLF_FORCE_DEVICE') is not None:
        n = os.environ.get('WORLD_SIZE')
    try:
        return max(1, int(n or 1))
    except ValueError:
        return 1


class PlacedBuffer(object):
    """A device buffer at a fixed virtual address made of `parts` equally sized physical chunks."""

    def __init__(self, backend, nbytes, align_offset=0, parts=None):
        self.backend = backend
        parts = parts or int(os.environ.get('SLF_PLACEMENT_PARTS', PARTS))
        gran = backend.vmm_granularity()
        need = int(nbytes) + 256
        self.parts = parts
        self.part_bytes = (need + parts * gran - 1) // (parts * gran) * gran
        self.total = self.part_bytes * parts
        self.va = backend.vmm_reserve(self.total)
        self.addr = self.va + int(align_offset)
        self.nbytes = int(nbytes)
        self.mapped = [None] * parts

    def map_part(self, i, handle):
        assert self.mapped[i] is None
        self.backend.vmm_map(self.va + i * self.part_bytes, self.part_bytes, handle)
        self.mapped[i] = handle

    def release(self):
        """Unmaps and releases the chunks and the address range."""
        b = self.backend
        for i, h in enumerate(self.mapped):
            if h is not None:
                b.vmm_unmap(self.va + i * self.part_bytes, self.part_bytes)
                b.vmm_chunk_release(h)
                self.mapped[i] = None
        if self.va:
            b.vmm_release_range(self.va, self.total)
            self.va = 0


_held = None        # spacer chunks kept alive by holding(): [(backend, handle)]


_held_count = 1     # placements expected inside the holding() block: each takes its share of the free memory for its spacers
_held_budget = None  # ... that share in bytes, fixed by the first placement of the block


class holding(object):
    """`with placement.holding(count):` around the set-up of SEVERAL simulations on one GPU in one process (the subdomain
    runners of a same-process group).  The spacers of every placement inside stay allocated until the block ends, so
    that the next placement starts behind them.  Without this the later arrays fall into the holes the earlier
    spacers left: a 0.2 GiB chunk fits sixteen times into the first 4 GiB hole, and the array is not spread at all.
    `count` = how many simulations will place arrays inside the block: every one takes 1 / count of the span and of the
    free memory for its spacers -- eight subdomains of BASELINE config 4 in one process held 8 x 70 GiB of spacers
    otherwise and the first plain allocation after them failed (round 5, `tools/bench_configs.py --only 3g8`)."""

    def __init__(self, count=1):
        self.count = max(1, int(count))

    def __enter__(self):
        global _held, _held_count, _held_budget
        self._outer = _held
        self._outer_count = _held_count
        if _held is None:
            _held = []
            _held_count = self.count
            _held_budget = None
        return self

    def __exit__(self, *exc):
        global _held, _held_count, _held_budget
        if self._outer is None:
            _held_count, _held_budget = self._outer_count, None
            held, _held = _held, None
            for backend, h in held:
                try:
                    backend.vmm_chunk_release(h)
                except Exception:  # noqa: BLE001 -- a backend that was closed in between took its context along
                    pass
        return False


def holding_now():
    """Inside a holding() block (several simulations being set up on one GPU)?"""
    return _held is not None


def place(backend, buffers, span=None):
    """Backs `buffers` (PlacedBuffer with equal part counts) with physical chunks spread over `span` bytes: part i of
    every buffer, then a spacer, for i = 0 .. parts-1; the spacers are released at the end (or when the enclosing
    holding() block ends).  Call it before anything else is allocated for the simulation: the spacers steer the
    allocator only while it hands out fresh memory in order (measured: 3-8 % when the macroscopic fields and the node
    map were allocated first, profiles/r02/README.md)."""
    parts = buffers[0].parts
    assert all(b.parts == parts for b in buffers)
    gran = backend.vmm_granularity()
    payload = sum(b.total for b in buffers)
    if span is None:
        span = int(os.environ.get('SLF_PLACEMENT_SPAN_GIB', SPAN >> 30)) << 30
    n = sharers()
    free = backend.free_memory() // n - payload     # what the device has left (this process's share of it), whoever holds the rest
    span = max(payload, min(span // n, payload + int((0.8 if n == 1 else 0.5) * max(0, free))))
    global _held_budget
    if _held is not None and _held_count > 1:
        # spacers stay allocated until the last simulation of the group is placed: a share of the memory for each
        if _held_budget is None:
            _held_budget = int(0.6 * max(0, backend.free_memory() // n)) // _held_count
        span = max(payload, min(span, payload + _held_budget))
    spacer = max(0, (span - payload) // parts) // gran * gran
    spacers = []
    try:
        for i in range(parts):
            for b in buffers:
                h = backend.vmm_chunk_create(b.part_bytes)
                try:
                    b.map_part(i, h)
                except Exception:
                    backend.vmm_chunk_release(h)        # not yet in b.mapped: nobody else would give it back
                    raise
            if spacer and i + 1 < parts:
                spacers.append(backend.vmm_chunk_create(spacer))
    finally:
        if _held is not None:
            _held.extend((backend, h) for h in spacers)
        else:
            for h in spacers:
                backend.vmm_chunk_release(h)
    return {'parts': parts, 'part_gib': round(buffers[0].part_bytes / 2.0 ** 30, 3),
            'spacer_gib': round(spacer / 2.0 ** 30, 3), 'span_gib': round((payload + spacer * (parts - 1)) / 2.0 ** 30, 1)}


def choose(make_set, measure, release, attempts=3, agree=0.02, log=None, room=None):
    """Placement by measurement.  The rule of place() -- spread the chunks over the span -- removes the 15 % bimodality
    of large arrays, but what a placement is worth still varies between processes and boxes (arrays of 1-2 GB: 35-41
    GMLUPS, profiles/r02/runner_path_placement.log).  So: make_set() places a set of arrays, measure(set) times the
    many-stream sweep on it (seconds per step); a second set is placed while the first is still allocated -- so it
    lands elsewhere -- and so on until two sets agree with the best within `agree` or `attempts` are used up.  The
    best set is returned, the others are released.  room() (optional) says whether another set fits next to the ones
    held; a further set that cannot be placed or probed ends the search with the best so far.  (Swapping the physical chunks under ONE address range instead was
    tried and is not safe on this stack: kernels launched after hipMemUnmap / hipMemMap of a range they had used before
    produced non-finite values -- profiles/r03/placement_remap_failure.txt.)"""
    sets, times, note = [], [], None
    for n in range(attempts):
        if n > 0 and room is not None and not room():
            note = 'no room for another set'
            break
        try:
            bufs = make_set()
        except (RuntimeError, MemoryError, OSError) as e:   # the backend's errors (HIPFatalError is a RuntimeError): a
            # further set that does not fit must not end a run that fitted before.  make_set() is all-or-nothing --
            # alloc_placed() gives back what it had placed before it re-raises or falls back -- so nothing is held here
            if n == 0:
                raise
            note = 'placing set %d failed: %s' % (n, str(e)[:100])
            break
        try:
            t = measure(bufs)
        except (RuntimeError, MemoryError, OSError) as e:
            release(bufs)
            if n == 0:
                raise
            note = 'probing set %d failed: %s' % (n, str(e)[:100])
            break
        sets.append(bufs)
        times.append(t)
        close = sorted(times)[:2]
        if len(times) > 1 and close[1] <= close[0] * (1.0 + agree):
            break
    best = min(range(len(times)), key=lambda i: times[i])
    for i, bufs in enumerate(sets):
        if i != best:
            release(bufs)
    info = {'times_ms': [round(x * 1e3, 4) for x in times], 'chosen': best}
    if note:
        info['note'] = note
    if log:
        log('placement by measurement: %s' % info)
    return sets[best], info


def probe_sweep(backend, desc, dim, src, dst, nbytes, stream, steps=12):
    """Seconds per step of the plain fluid sweep (same lattice, precision, sizes and access pattern as the module
    described by `desc`; BGK, every axis wrapped in-sweep, no node map) between the distribution arrays at `src` and
    `dst` (the same address for the AA pattern): the many-stream access pattern whose speed depends on the placement.
    The arrays are scratch: filled with one positive value, i.e. a fluid at rest."""
    from sailfish_amd import hipabi
    b = backend
    d = hipabi.SlfModuleDesc.from_buffer_copy(desc)
    d.model, d.simtype, d.fluid_only, d.n_types, d.has_force = hipabi.SLF_BGK, 0, 1, 0, 0
    d.incompressible, d.node_addressing, d.relaxation_enabled, d.tau = 0, 0, 1, 1.0
    for a in range(3):
        d.periodic_fused[a] = d.periodic_local[a] = int(a < dim)
    ab = src != dst
    d.access_pattern = hipabi.SLF_AB if ab else hipabi.SLF_AA
    module = b.build(d)
    for addr in set((src, dst)):
        b.memset_buf(addr, 0x3D, nbytes, stream)        # 0x3d3d3d3d = 0.046 as a float, 4.2e-14 as a double
    sig = 'P' * (4 + dim) + 'i'
    pairs = [(src, dst), (dst, src)] if ab else [(src, src)]
    ks = [b.get_kernel(module, 'CollideAndPropagate', (64,), [0, i, o, i] + [i] * dim + [0], sig,
                       needs_iteration=not ab) for i, o in pairs]
    ev0 = None
    for it in range(4 + steps):
        if it == 4:
            ev0 = b.make_event(stream, timing=True)
        b.set_iteration(it)
        b.run_kernel(ks[it % len(ks)], None, stream)
    ev1 = b.make_event(stream, timing=True)
    ev1.synchronize()
    b.set_iteration(0)
    seconds = ev1.time_since(ev0) * 1e-3 / steps
    for addr in set((src, dst)):                        # back to what alloc_placed() handed out: zeros
        b.memset_buf(addr, 0, nbytes, stream)
    stream.synchronize()
    del ks, module                                      # the probe's module and kernels go with it
    return seconds


def room_for(backend, nbytes, slack=1.15):
    """True when another set of `nbytes` of distribution arrays fits into the device memory that is free right now."""
    try:
        return backend.free_memory() // sharers() >= int(nbytes * slack)
    except Exception:  # noqa: BLE001
        return False


def _check(lib, status, what):
    if status != 0:
        from sailfish_amd.backend_hip import HIPFatalError
        msg = lib.slf_last_error()
        raise HIPFatalError('%s failed (status %d): %s' % (what, status, msg.decode() if msg else '?'))


class VmmMixin(object):
    """Thin ctypes wrappers of the slf_vmm_* entry points (mixed into HIPBackend)."""

    def free_memory(self):
        n = ctypes.c_size_t()
        _check(self._lib, self._lib.slf_ctx_free_memory(self._ctx, ctypes.byref(n)), 'slf_ctx_free_memory')
        return int(n.value)

    def vmm_granularity(self):
        n = ctypes.c_size_t()
        _check(self._lib, self._lib.slf_vmm_granularity(self._ctx, ctypes.byref(n)), 'slf_vmm_granularity')
        return int(n.value)

    def vmm_reserve(self, nbytes):
        p = ctypes.c_void_p()
        _check(self._lib, self._lib.slf_vmm_reserve(self._ctx, int(nbytes), ctypes.byref(p)), 'slf_vmm_reserve')
        return p.value

    def vmm_release_range(self, va, nbytes):
        _check(self._lib, self._lib.slf_vmm_release_range(self._ctx, ctypes.c_void_p(va), int(nbytes)),
               'slf_vmm_release_range')

    def vmm_chunk_create(self, nbytes):
        h = ctypes.c_uint64()
        _check(self._lib, self._lib.slf_vmm_chunk_create(self._ctx, int(nbytes), ctypes.byref(h)), 'slf_vmm_chunk_create')
        return int(h.value)

    def vmm_chunk_release(self, handle):
        _check(self._lib, self._lib.slf_vmm_chunk_release(self._ctx, ctypes.c_uint64(handle)), 'slf_vmm_chunk_release')

    def vmm_map(self, va, nbytes, handle):
        _check(self._lib, self._lib.slf_vmm_map(self._ctx, ctypes.c_void_p(va), int(nbytes), ctypes.c_uint64(handle)),
               'slf_vmm_map')

    def vmm_unmap(self, va, nbytes):
        _check(self._lib, self._lib.slf_vmm_unmap(self._ctx, ctypes.c_void_p(va), int(nbytes)), 'slf_vmm_unmap')
