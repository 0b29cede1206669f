"""x faces connected to another subdomain without ghost-column traffic (C ABI: slf_module_set_xface_buffers).

A workgroup owns whole rows, so an x face is one node per row: pushing into the ghost columns costs a partial-line
write per row and direction, packing / unpacking them a strided gather / scatter (64-byte sectors for 4-byte values).
Here the two edge lanes of every row write what leaves the subdomain into dense send buffers and read what enters it
from dense receive buffers ([z][k][y] over the padded plane, k = rank of the direction among the 5 with e_x > 0 resp.
< 0: a range of z-planes is one contiguous piece); per step the host moves send_high -> the high neighbour's recv_low
and send_low -> the low neighbour's recv_high, nothing else.  The distribution arrays are then stale at the face until
`materialise()` copies the receive buffers into them (before a checkpoint / a debug dump); receive buffers whose bits
are all ones (memset 0xFF) mean "the arrays count" -- the marker is compared bit for bit, so a NaN or an infinity a
diverging neighbour really sent crosses the face like any other value.

Overlap (the reference puts the x-face blocks into its boundary kernel and overlaps their transfer with the bulk,
subdomain_runner.py:409-419, 1028-1058; a whole-row workgroup cannot split off an x face): the sweep is cut into
z-chunks, one launch each (`ChunkPlan`; the chip drains at a launch boundary, which costs 1-2 % at 128 x 512 x 512 with
four chunks); as soon as the chunks that write a range of face-buffer planes are done that range travels on the halo
stream while the next chunk computes, and a chunk of the NEXT step waits only for the planes it reads.  Send and
receive buffers exist twice and alternate by step parity, so a transfer never races with the following step's writes.
(One launch that signals the completion of its chunks from inside the kernel was tried: the release fence a workgroup
needs before it may count itself writes the L2 of its XCD back, 25 x slower -- profiles/r03/xface_overlap_schemes.jsonl.)

Replaces, for 1-D decompositions along x (the reference's default axis, geo.py:100-135), the reference's
CollectContinuousData / DistributeContinuousData on x faces (kernel_utils.mako:526-543, 692-708).
"""
import os

import numpy as np

from sailfish_amd import sym

LOW, HIGH = 0, 1
NXD = 5          # D3Q19 directions per sign of e_x


def supported(grid, desc, indirect=False, simtype=0):
    """Can a module built from `desc` take x-face buffers (slf_module_set_xface_buffers)?  D3Q19 single fluid, direct
    addressing, rows of at most 1024 nodes -- and the whole-row kernels must be what runs: the --minimize_roundoff
    formulation lives in the per-node kernels only (slf_api.hip: variant = 0), and so does everything under an
    SLF_VARIANT without bit 8."""
    from sailfish_amd import hipabi
    if int(desc.incompressible) == hipabi.SLF_DENSITY_ROUNDOFF or int(desc.regularized) or int(desc.subgrid):
        return False
    variant = os.environ.get('SLF_VARIANT')
    if variant is not None and not (int(variant) & 8):
        return False
    return grid.dim == 3 and grid.Q == 19 and not indirect and not simtype and desc.lat_nx - 2 <= 1024


def face_count(desc):
    """Elements of one face buffer: the padded (arr_ny x arr_nz) plane x 5 directions."""
    return NXD * desc.arr_ny * desc.arr_nz


class ChunkPlan(object):
    """Pure host logic: the z-chunks of the sweep of an x-connected subdomain, the order they are swept in, which
    planes of the face buffers may travel after each, and which transfer a chunk of the following step waits for.

    Step kinds: 'push' (AB, odd AA step): a node in plane z writes the send-buffer planes z-1 .. z+1 (the row its
    target sits in) and the following step reads a node's own plane; 'own' (even AA step): writes the own plane,
    and the following (odd) step pulls from the planes z-1 .. z+1.
    """

    def __init__(self, nz, wrap_z, nchunks=None, min_planes=8):
        if nchunks is None:
            nchunks = int(os.environ.get('SLF_XFACE_CHUNKS', '4'))
        k = max(1, min(int(nchunks), nz // max(1, min_planes)))
        self.nz, self.wrap_z = nz, bool(wrap_z)
        bounds = [1 + (nz * i) // k for i in range(k + 1)]
        self.chunks = [(bounds[i], bounds[i + 1]) for i in range(k)]        # real planes [z0, z1)
        # z wrapped in-sweep: the chunks form a ring; the step must not END with a neighbour of the chunk the next
        # step STARTS with (its planes would still be travelling)
        self.order = [0, k - 1] + list(range(1, k - 1)) if (self.wrap_z and k >= 4) else list(range(k))
        self.batches, self.need = {}, {}
        for kind in ('push', 'own'):
            self.batches[kind], self.need[kind] = self._plan(kind)

    def _wrap(self, p):
        if self.wrap_z:
            return ((p - 1) % self.nz) + 1
        return p

    def writes(self, kind, c):
        a, b = self.chunks[c]
        ps = range(a - 1, b + 1) if kind == 'push' else range(a, b)
        return sorted(set(self._wrap(p) for p in ps))

    def reads_after(self, kind, c):
        a, b = self.chunks[c]
        ps = range(a, b) if kind == 'push' else range(a - 1, b + 1)
        return sorted(set(self._wrap(p) for p in ps))

    def _plan(self, kind):
        done_at = {}                               # plane -> position in the order after which it is complete
        for pos, c in enumerate(self.order):
            for p in self.writes(kind, c):
                done_at[p] = pos                   # the LAST writer decides
        # SLF_XFACE_BATCHES=two: only two transfers per step -- everything that is complete before the last chunk starts
        # travels while the last chunk is swept, the rest while the first chunk of the next step is swept (which, by the
        # order above, reads none of it): two event records and two RCCL groups less per step.  Measured SLOWER with one
        # rank sending to itself (0.935 against 0.913 ms per step, AB 1.02 against 0.98: the three-quarter-face transfer
        # is a 200 us RCCL self-copy beside the last chunk, and the next step waits for it; profiles/NOTES.md), so one
        # transfer after every chunk stays the default; over xGMI the balance may differ.
        if os.environ.get('SLF_XFACE_BATCHES', 'all') == 'two':
            first = max(0, len(self.order) - 2)
            for p in done_at:
                done_at[p] = max(done_at[p], first)
        batches = []
        for pos in range(len(self.order)):
            planes = sorted(p for p, d in done_at.items() if d == pos)
            runs = []
            for p in planes:
                if runs and runs[-1][1] == p:
                    runs[-1][1] = p + 1
                else:
                    runs.append([p, p + 1])
            batches.append([tuple(r) for r in runs])
        need = []
        for c in range(len(self.chunks)):
            js = [done_at[p] for p in self.reads_after(kind, c) if p in done_at]
            need.append(max(js) if js else -1)
        return batches, need

    def peer_need(self, kind, prev_kind):
        """`need` for face buffers that are NOT copied: my send planes ARE the neighbour's receive planes (peer transport
        between processes, sailfish_amd/peer.py; subdomains of one process, controller.LocalGroup).  Chunk c of a step of
        kind `kind` then waits for the position of the neighbours' PREVIOUS step after which (a) the planes it reads are
        complete -- need[prev_kind][c], as with copies -- and (b) the planes it WRITES are no longer being read: the
        neighbours' previous step read the set this step writes (the sets alternate by step parity), chunk c' of it the
        planes reads_after(kind, c') -- what the step before, of this step's kind, had left there.  Both sides run the
        same plan (1-D x decomposition: equal y / z extents, same chunk count), and a position's signal follows the
        completion of its chunk, so one position per chunk covers both.  Needs a signal after EVERY position
        (exchanges_at is not consulted by the callers in this mode)."""
        pos_of = dict((c, pos) for pos, c in enumerate(self.order))
        out = []
        for c in range(len(self.chunks)):
            mine = set(self.writes(kind, c))
            war = [pos_of[c2] for c2 in range(len(self.chunks)) if mine & set(self.reads_after(kind, c2))]
            out.append(max([self.need[prev_kind][c]] + war))
        return out

    def peer_signals(self, kind, next_kind):
        """Positions of a step of kind `kind` after which the neighbours need a signal: those some chunk of THEIR next
        step (kind `next_kind`) waits for.  Signals are counted, so a wait for position p is a wait for the signals up to
        and including p -- peer_counts() turns positions into counts."""
        return sorted(set(self.peer_need(next_kind, kind)))

    def peer_counts(self, kind, prev_kind):
        """[(chunk position, signals of the neighbours' previous step to wait for before it)] for a step of kind `kind`
        that follows a step of kind `prev_kind`: the first chunk whose need is a signalled position waits for every
        signal up to that position that no earlier chunk has waited for.  The counts add up to len(peer_signals(prev_kind,
        kind)): a step consumes exactly what the step before produced."""
        signalled = self.peer_signals(prev_kind, kind)
        need = self.peer_need(kind, prev_kind)
        out, consumed = [], 0
        for pos, c in enumerate(self.order):
            upto = sum(1 for p in signalled if p <= need[c])
            if upto > consumed:
                out.append((pos, upto - consumed))
                consumed = upto
        assert consumed == len(signalled)
        return out

    def exchanges_at(self, pos):
        """Does a transfer (of either step kind) start after the chunk at position `pos`?  Positions without one need no
        event either: nothing waits for them."""
        return any(self.batches[kind][pos] for kind in self.batches)

    def region(self, c, ny):
        z0, z1 = self.chunks[c]
        return (1, ny + 1, z0, z1)

    def neighbours(self, c):
        """Chunks whose sweep touches planes that chunk c reads or writes (itself and the chunks next to it; around the
        ring when z is wrapped in-sweep): what a chunk of the next step has to wait for when the chunks do not all
        run on one stream."""
        k = len(self.chunks)
        out = []
        for d in (-1, 0, 1):
            n = c + d
            if self.wrap_z:
                n %= k
            if 0 <= n < k and n not in out:
                out.append(n)
        return out


class XFaceHalo(object):
    def __init__(self, backend, module, grid, desc, send, recv, shared=False):
        """send / recv: [parity][face] device addresses of buffers of face_count(desc) reals each, 0 for a face that
        is not connected (or one [face] list: no alternation).  shared: the send buffers are a neighbour's receive
        buffers (its memory, mapped here: peer transport) -- this side never fills them.  Every buffer this subdomain
        owns starts out as 'nothing has crossed here' (reset(); the C ABI's contract, include/sailfish_hip.h)."""
        self.backend, self.module, self.grid, self.desc = backend, module, grid, desc
        self.dtype = np.float32 if desc.precision == 4 else np.float64
        self.plane = NXD * desc.arr_ny                  # elements of one z-plane of a face buffer
        self.count = face_count(desc)
        self.nbytes = self.count * self.dtype().itemsize
        if not isinstance(send[0], (list, tuple)):
            send, recv = [send, send], [recv, recv]
        self.send, self.recv = [list(s) for s in send], [list(r) for r in recv]
        # directions entering through the low face have e_x > 0, through the high face e_x < 0 (ascending = rank order)
        self.enter = [sym.get_prop_dists(grid, 1, 0), sym.get_prop_dists(grid, -1, 0)]
        self._kernels = {}
        self._bound = None
        self.shared = bool(shared)
        self.bind(0, 1)
        self.reset()

    @classmethod
    def allocate(cls, backend, module, grid, desc, faces, alloc):
        """faces: (low connected, high connected); alloc(n_elements) -> device address of a buffer the transport can
        send from / receive into.  Allocation order, per parity: send low, send high, receive low, receive high."""
        n = face_count(desc)
        send, recv = [], []
        for _ in range(2):
            send.append([alloc(n) if faces[f] else 0 for f in (LOW, HIGH)])
            recv.append([alloc(n) if faces[f] else 0 for f in (LOW, HIGH)])
        return cls(backend, module, grid, desc, send, recv)

    def bind(self, send_parity, recv_parity):
        """The sweeps launched from now on write send[send_parity] and read recv[recv_parity]."""
        key = (send_parity, recv_parity)
        if key == self._bound:
            return
        s, r = self.send[send_parity], self.recv[recv_parity]
        self.backend.set_xface_buffers(self.module, s[LOW], s[HIGH], r[LOW], r[HIGH])
        self._bound = key

    def begin_step(self, iteration, stream):
        """Buffers of the step that starts at `iteration`: it writes the send set of its parity and reads what the
        previous step's transfers delivered."""
        par = iteration & 1
        self.bind(par, 1 - par)
        if self.needs_clear:
            self.clear_send(stream, par)
        return par

    def _all(self):
        seen = []
        for group in self.send + self.recv:
            for a in group:
                if a and a not in seen:
                    seen.append(a)
        return seen

    shared = False      # my send buffers ARE the neighbours' receive buffers (same-process groups: controller.LocalGroup)

    def reset(self, stream=None):
        """All bits set: nothing has crossed the faces yet, the kernels read the arrays.  With shared buffers only what
        this subdomain RECEIVES is cleared: its send buffers are a neighbour's input (still valid for that neighbour), and
        the next step rewrites every entry of them that is ever written."""
        bufs = self._all()
        if self.shared:
            bufs = []
            for group in self.recv:
                for a in group:
                    if a and a not in bufs:
                        bufs.append(a)
        for a in bufs:
            self.backend.memset_buf(a, 0xFF, self.nbytes, stream)

    @property
    def needs_clear(self):
        """Clear the send buffers before every step?  No (round 5).  Entries a step does not write -- rows whose source
        would be a ghost row (y / z not wrapped in-sweep), edge nodes that the node map excludes -- must read as 'nothing
        crossed here'; rounds 3-4 cleared the buffers every step for that (two fills of the whole face per step and
        face).  But WHICH entries a step writes is a property of the geometry, not of the step: the node map is static,
        every row is swept every step, and a send set only ever sees one kind of step (in place: the even steps own set
        0, the odd ones set 1; two-copy: every step pushes) -- so an entry that is not written now was never written
        and still holds the all-ones of reset(), which every host-side write of the state repeats.  Traced on a pipe cut
        into three x-slabs in one process, the fills were 16 % of the GPU's time (profiles/r05/kernel_stats_pipe_3x_before.csv).
        SLF_XFACE_CLEAR=1 brings the per-step clear back (A/B)."""
        if os.environ.get('SLF_XFACE_CLEAR', '0') != '1':
            return False
        d = self.desc
        return not (d.periodic_fused[1] and d.periodic_fused[2]) or not d.fluid_only

    def clear_send(self, stream, parity=0):
        for a in self.send[parity]:
            if a:
                self.backend.memset_buf(a, 0xFF, self.nbytes, stream)

    def prime_pull(self, dist, stream, parity=0):
        """The inverse of materialise(pushed=False): copies the ghost columns of `dist` (opposite slots -- what the odd
        in-place step pulls across a connected face) into the receive buffers of `parity`.  Needed when a state is
        written from the host at an odd in-place iteration (checkpoint restore, set_dist): rows whose edge lanes skip
        the pull out of the ghost column (fluid-only row kernels, slf_row.hip SKIP_GHOST_PULL) rely on the receive
        buffers alone."""
        self._ghost_columns('CollectContinuousData', dist, stream, parity)

    def _ghost_columns(self, name, dist, stream, parity):
        b, d = self.backend, self.desc
        nx = d.lat_nx - 2
        isz = self.dtype().itemsize
        recv = self.recv[parity]
        for face in (LOW, HIGH):
            if not recv[face]:
                continue
            x = 0 if face == LOW else nx + 1
            for k, q in enumerate(self.enter[face]):
                key = (name, dist, recv[face], k)
                if key not in self._kernels:
                    self._kernels[key] = b.get_kernel(self.module, name, (64,),
                                                      [dist, recv[face] + k * d.arr_ny * isz, 1 << self.grid.idx_opposite[q],
                                                       x, d.arr_nx, d.arr_ny, d.arr_nx * d.arr_ny, d.arr_nz, d.arr_ny,
                                                       NXD * d.arr_ny], 'PPiiiiiiii')
                b.run_kernel(self._kernels[key], None, stream)

    def materialise(self, dist, pushed, stream, parity=0):
        """Writes the receive buffers of `parity` into the distribution array `dist`: pushed = True after a push step
        (AB, odd AA: the values belong into the first real column, same slots), False after the even AA step (they
        belong into the ghost column, opposite slots, where the next pull looks for them)."""
        b, d = self.backend, self.desc
        nx = d.lat_nx - 2
        isz = self.dtype().itemsize
        recv = self.recv[parity]
        for face in (LOW, HIGH):
            if not recv[face]:
                continue
            if pushed:
                x = 1 if face == LOW else nx
                mask = 0
                for q in self.enter[face]:
                    mask |= 1 << q
                jobs = [(mask, x, 0)]
            else:
                x = 0 if face == LOW else nx + 1
                jobs = [(1 << self.grid.idx_opposite[q], x, k * d.arr_ny * isz) for k, q in enumerate(self.enter[face])]
            for mask, col, off in jobs:
                key = (dist, recv[face], pushed, mask)
                if key not in self._kernels:
                    # node box: columns = y (stride arr_nx), rows = z; buffer [z][k][y]: directions arr_ny apart,
                    # rows 5 arr_ny apart
                    self._kernels[key] = b.get_kernel(self.module, 'DistributeContinuousData', (64,),
                                                      [dist, recv[face] + off, mask, col, d.arr_nx, d.arr_ny,
                                                       d.arr_nx * d.arr_ny, d.arr_nz, d.arr_ny, NXD * d.arr_ny],
                                                      'PPiiiiiiii')
                b.run_kernel(self._kernels[key], None, stream)


# ---- binary Shan-Chen over connected x faces (C ABI: slf_module_set_xface_planes) ---------------------------------------
NN_FIELDS = 2    # rho, phi


def supported_nn(grid, desc, indirect=False):
    """Can a Shan-Chen module (binary or single-component) built from `desc` take x-face planes?  What
    slf_module_set_xface_planes checks: D3Q19, direct addressing, rows of 2 .. 1024 nodes, whole-row kernels, x not wrapped
    inside the kernels; both access patterns, with or without a node map.  (Which rows of a density plane an edge node may
    read is the runner's check: NNSubdomainRunner._nn_x_faces_only.)"""
    from sailfish_amd import hipabi
    if int(desc.simtype) not in (hipabi.SLF_SIM_SHAN_CHEN_BINARY, hipabi.SLF_SIM_SHAN_CHEN_SINGLE):
        return False
    variant = os.environ.get('SLF_VARIANT')
    if variant is not None and not (int(variant) & 8):
        return False
    if os.environ.get('SLF_SC_XFACE', '1') == '0':
        return False
    if int(desc.simtype) == hipabi.SLF_SIM_SHAN_CHEN_BINARY and os.environ.get('SLF_SC_FUSED', '2') != '2':
        return False        # binary: the planes are served by ShanChenPrepareDensities / ShanChenCollideAndPropagateFusedV only
    nx = desc.lat_nx - 2
    return grid.dim == 3 and grid.Q == 19 and not indirect and 2 <= nx <= 1024 and not desc.periodic_fused[0]


class NNPlanes(object):
    """The three sets of planes of one subdomain: populations of lattice 0 and 1 (face_count(desc) reals per face each) and
    the densities (NN_FIELDS * arr_ny * arr_nz).  A neighbour link moves ONE buffer per kind and step parity:
    'dist' = [face][lattice][planes], 'macro' = [face][planes]; send[kind][parity][face] / recv[...] are the addresses of a
    face's part.  Set p of the population planes is written by the sweep of the steps of parity p and read by the two
    kernels of the following step; set p of the density planes is written and read within the steps of parity p."""

    def __init__(self, backend, module, grid, desc, n_lat=2):
        self.backend, self.module, self.grid, self.desc = backend, module, grid, desc
        self.dtype = np.float32 if desc.precision == 4 else np.float64
        self.isz = self.dtype().itemsize
        self.n_lat = int(n_lat)                              # 2: binary model; 1: single component (field 0 of the densities)
        self.n_dist = face_count(desc)                       # one lattice, one face
        self.n_macro = NN_FIELDS * desc.arr_ny * desc.arr_nz
        self.count = {'dist': self.n_lat * self.n_dist, 'macro': self.n_macro}      # elements per face and kind
        self.send = dict((k, [[0, 0], [0, 0]]) for k in self.count)
        self.recv = dict((k, [[0, 0], [0, 0]]) for k in self.count)
        self.shared = False
        self.enter = [sym.get_prop_dists(grid, 1, 0), sym.get_prop_dists(grid, -1, 0)]
        self._kernels = {}

    def program_bind(self, q, it):
        """The kernels launched from here on serve step `it`: they write the population planes of its parity, read those
        of the other one, and exchange densities through the planes of its parity."""
        par = it & 1
        ds, dr = self.send['dist'][par], self.recv['dist'][1 - par]
        off = self.n_dist * self.isz
        for lat in range(self.n_lat):
            q.xface_planes(self.module, lat, *[a + lat * off if a else 0 for a in (ds[LOW], ds[HIGH], dr[LOW], dr[HIGH])])
        ms, mr = self.send['macro'][par], self.recv['macro'][par]
        q.xface_planes(self.module, 2, ms[LOW], ms[HIGH], mr[LOW], mr[HIGH])

    def _buffers(self, which, kind):
        out = []
        for par in (0, 1):
            for f in (LOW, HIGH):
                a = which[kind][par][f]
                if a and (a, self.count[kind] * self.isz) not in out:
                    out.append((a, self.count[kind] * self.isz))
        return out

    def reset(self, stream=None):
        """Population planes, all bits set: nothing has crossed the faces, the arrays count (state written from the host).
        Shared planes: only what this subdomain receives (its send planes are a neighbour's input and are cleared by that
        neighbour).  The density planes carry no markers: prime() defines them."""
        bufs = self._buffers(self.recv, 'dist')
        if not self.shared:
            bufs = bufs + [b for b in self._buffers(self.send, 'dist') if b not in bufs]
        for a, nbytes in bufs:
            self.backend.memset_buf(a, 0xFF, nbytes, stream)

    def prime(self, fields, stream=None):
        """Every entry of MY density send planes (both parities) from the first / last real column of `fields` (device
        addresses of rho [, phi]) as they are now: the densities of nodes the density pass never rewrites (walls, ghost
        rows) -- what the neighbour's ghost column would hold -- and a defined start for the others.  After the state was
        written from the host, and once the send planes are final (shared planes: after the neighbours' planes were
        adopted)."""
        b, d = self.backend, self.desc
        nx = d.lat_nx - 2
        for par in (0, 1):
            for face in (LOW, HIGH):
                plane = self.send['macro'][par][face]
                if not plane:
                    continue
                x = 1 if face == LOW else nx
                for j, field in enumerate(fields):
                    key = ('prime', field, plane, j)
                    if key not in self._kernels:
                        # node box: columns = y (stride arr_nx), rows = z; plane [z][field][y]
                        self._kernels[key] = b.get_kernel(self.module, 'CollectContinuousData', (64,),
                                                          [field, plane + j * d.arr_ny * self.isz, 1, x, d.arr_nx, d.arr_ny,
                                                           d.arr_nx * d.arr_ny, d.arr_nz, d.arr_ny, NN_FIELDS * d.arr_ny],
                                                          'PPiiiiiiii')
                    b.run_kernel(self._kernels[key], None, stream)

    def prime_own(self, dists, stream=None):
        """In-place pattern: the population planes of the EVEN steps (set 0) from my first / last real column as it is
        now -- entry (row, direction I) = slot opp(I) of the edge node of that row, which is where the even step leaves
        what it also stores into the plane.  For the entries no step ever writes: an edge node the node map excludes never
        sends, and the neighbour's odd step would fall back to its ghost column, which nothing maintains here (a wall node
        of the single-component model, whose density the pass in front forms too, pulls from such a node: an infinity out
        of the ghost column would end up in its density and from there in the force on the fluid next to it).  With ghost
        columns the exchange delivers exactly these values, unchanged, every step."""
        b, d = self.backend, self.desc
        nx = d.lat_nx - 2
        for face in (LOW, HIGH):
            plane = self.send['dist'][0][face]
            if not plane:
                continue
            x = 1 if face == LOW else nx
            leaving = self.enter[HIGH] if face == LOW else self.enter[LOW]      # e_x < 0 leave through the low face
            for lat, dist in enumerate(dists):
                for k, q in enumerate(leaving):
                    dst = plane + lat * self.n_dist * self.isz + k * d.arr_ny * self.isz
                    key = ('own', dist, dst, q)
                    if key not in self._kernels:
                        self._kernels[key] = b.get_kernel(self.module, 'CollectContinuousData', (64,),
                                                          [dist, dst, 1 << self.grid.idx_opposite[q], x, d.arr_nx, d.arr_ny,
                                                           d.arr_nx * d.arr_ny, d.arr_nz, d.arr_ny, NXD * d.arr_ny],
                                                          'PPiiiiiiii')
                    b.run_kernel(self._kernels[key], None, stream)

    def materialise(self, dists, pushed, stream, parity):
        """Writes the population planes received in the steps of `parity` into the arrays `dists` = (lattice 0, lattice 1)
        those steps wrote -- before anything reads the arrays on the host (checkpoint, debug dump).  pushed = True after a
        push step (two-copy, odd in-place): the values belong into the first / last real column, same slots; False after
        the even in-place step: into the ghost column, opposite slots, where the next pull looks for them (the kernels pull
        from the ghost columns wherever a plane says 'nothing crossed here', i.e. after the reset that follows a restore)."""
        b, d = self.backend, self.desc
        nx = d.lat_nx - 2
        recv = self.recv['dist'][parity]
        for face in (LOW, HIGH):
            if not recv[face]:
                continue
            if pushed:
                x = 1 if face == LOW else nx
                mask = 0
                for q in self.enter[face]:
                    mask |= 1 << q
                jobs = [(mask, x, 0)]
            else:
                x = 0 if face == LOW else nx + 1
                jobs = [(1 << self.grid.idx_opposite[q], x, k * d.arr_ny * self.isz) for k, q in enumerate(self.enter[face])]
            for lat, dist in enumerate(dists):
                for mask, col, off in jobs:
                    src = recv[face] + lat * self.n_dist * self.isz + off
                    key = (dist, src, mask, col)
                    if key not in self._kernels:
                        self._kernels[key] = b.get_kernel(self.module, 'DistributeContinuousData', (64,),
                                                          [dist, src, mask, col, d.arr_nx, d.arr_ny, d.arr_nx * d.arr_ny,
                                                           d.arr_nz, d.arr_ny, NXD * d.arr_ny], 'PPiiiiiiii')
                    b.run_kernel(self._kernels[key], None, stream)
