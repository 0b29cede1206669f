"""Full-size parity without a full-size oracle run: z-plane windows.

TEST INFRASTRUCTURE ONLY (see lbm_oracle.c): used by tests/ and by bench.py's validation leg after the timed
region, never by sailfish_amd/.

The table-driven oracle does ~0.5 MLUPS per core; a 512^3 box is out of its reach in a test, but the state of one
z-plane after two steps depends only on the five planes around it (one push / pull = one plane per step; the
in-place AA pair "even (local) + odd (pull and push)" = two planes).  So: copy the seven planes z-3 .. z+3 of the
GPU's arrays into a 7-plane oracle subdomain that keeps the full x / y extent (node map, in-sweep wrap or
ghost-layer PBC along x and y, walls, lid -- everything as in the full run), advance both by two steps, and compare
plane z bit for bit.  Planes whose window would stick out of a non-periodic box are served by a window pushed
against the box's own ghost plane; a z axis wrapped in-sweep wraps the window.
"""
import numpy as np

from oracle.oracle import OracleSim
from sailfish_amd import hipabi

RADIUS = 2            # planes a plane's state depends on, per side, after two steps
W = 2 * (RADIUS + 1) + 1


def _clone_desc(desc, **changes):
    d = hipabi.SlfModuleDesc.from_buffer_copy(desc)
    for k, v in changes.items():
        setattr(d, k, v)
    return d


class PlaneWindow(object):
    """One sampled plane `z` (1-based real plane index) of a D3Q19 subdomain."""

    def __init__(self, desc, node_map, z):
        nz = desc.lat_nz - 2
        if desc.lat_nz < W:
            raise ValueError('subdomain too thin for a %d-plane window' % W)
        self.z = z
        if desc.periodic_fused[2]:
            self.planes = [((z - (RADIUS + 1) + k - 1) % nz) + 1 for k in range(W)]
            self.t = RADIUS + 1
        else:
            if desc.periodic_local[2] and (z - RADIUS < 1 or z + RADIUS > nz):
                raise ValueError('ghost-layer PBC along z: sample planes at least %d from the faces' % RADIUS)
            lo = min(max(z - (RADIUS + 1), 0), nz + 2 - W)
            self.planes = list(range(lo, lo + W))
            self.t = z - lo
        wd = _clone_desc(desc, lat_nz=W, arr_nz=W, dist_stride=0)
        wd.periodic_fused[2] = 0
        wd.periodic_local[2] = 0
        self.desc, self.full = wd, desc
        self.o = OracleSim(wd)
        self.map = None if node_map is None else np.ascontiguousarray(node_map[self.planes])
        self.aa = desc.access_pattern == hipabi.SLF_AA
        self.pbc_axes = [a for a in (0, 1) if desc.periodic_local[a] and not desc.periodic_fused[a]]
        self.dist = None
        self.off_y = self.off_x = 0          # where the subdomain's node (y, x) sits in the window's arrays

    def runs(self):
        """[(first global plane, count, first local plane)]: stretches of consecutive planes (one copy each)."""
        out, k = [], 0
        while k < W:
            j = k
            while j + 1 < W and self.planes[j + 1] == self.planes[j] + 1:
                j += 1
            out.append((self.planes[k], j - k + 1, k))
            k = j + 1
        return out


class PlaneCheck(object):
    """Seeds plane windows from a device-resident state, advances them with the oracle and compares the sampled
    planes with the device again.  `fetch(addr, byte_offset, nbytes) -> bytes-like` reads device memory."""

    def __init__(self, backend, desc, node_map, zs, dist_addrs, stride, field_addrs=None):
        self.backend, self.desc, self.stride = backend, desc, int(stride)
        self.dtype = np.float32 if desc.precision == 4 else np.float64
        self.isz = self.dtype().itemsize
        self.nxy = desc.arr_nx * desc.arr_ny
        self.dist_addrs = list(dist_addrs)
        self.field_addrs = field_addrs            # [rho, vx, vy, vz] or None
        self.windows = [PlaneWindow(desc, node_map, z) for z in zs]
        self.iteration = None

    # -- device access ------------------------------------------------------
    def _fetch_planes(self, base, first_plane, count):
        out = np.empty(count * self.nxy, dtype=self.dtype)
        self.backend.from_buf(base + first_plane * self.nxy * self.isz, out)
        return out.reshape(count, self.desc.arr_ny, self.desc.arr_nx)

    def _fetch_window(self, w, addr):
        d = np.empty((19, W, self.desc.arr_ny, self.desc.arr_nx), dtype=self.dtype)
        for q in range(19):
            base = addr + q * self.stride * self.isz
            for g0, n, k0 in w.runs():
                d[q, k0:k0 + n] = self._fetch_planes(base, g0, n)
        return d

    # -- protocol -------------------------------------------------------------
    def seed(self, iteration):
        """Call with the device idle; `iteration` = number of steps done so far (decides the AA parity / which
        copy is the current one)."""
        self.iteration = int(iteration)
        for w in self.windows:
            w.dist = [self._fetch_window(w, a) for a in self.dist_addrs]
            w.rho = np.full(w.o.shape, np.inf, dtype=self.dtype)
            w.v = [np.full(w.o.shape, np.inf, dtype=self.dtype) for _ in range(3)]

    def advance(self, steps=2, save_last=True):
        if steps > RADIUS:
            raise ValueError('a window is exact for at most %d steps' % RADIUS)
        for w in self.windows:
            region = (1, w.desc.lat_ny - 1, 1, W - 1)
            it = self.iteration
            for s in range(steps):
                opts = 1 if (save_last and s == steps - 1) else 0
                if w.aa:
                    prop = 2 if (it & 1) else 1
                    w.o.step(prop, w.map, w.dist[0], w.dist[0], w.rho, w.v[0], w.v[1], w.v[2], opts, region)
                    out, swap = 0, (it & 1) == 0
                else:
                    i = it & 1
                    w.o.step(0, w.map, w.dist[i], w.dist[1 - i], w.rho, w.v[0], w.v[1], w.v[2], opts, region)
                    out, swap = 1 - i, False
                for axis in w.pbc_axes:
                    w.o.pbc(w.dist[out], axis, swap)
                it += 1
        self.iteration += steps

    def compare(self, fields=True):
        """Device planes against the windows.  Returns {'planes', 'nodes', 'dist_exact', 'dist_err', 'rho_err',
        'v_abs_err'}; the sampled planes are compared over their real nodes (1..ny, 1..nx), every population."""
        d = self.desc
        ys, xs = slice(1, d.lat_ny - 1), slice(1, d.lat_nx - 1)
        cur = 0 if len(self.dist_addrs) == 1 else (self.iteration & 1)
        res = {'planes': [w.z for w in self.windows], 'nodes': 0, 'dist_exact': True, 'dist_err': 0.0,
               'rho_err': 0.0, 'v_abs_err': 0.0, 'compared_values': 0}
        for w in self.windows:
            wys = slice(ys.start + w.off_y, ys.stop + w.off_y)
            wxs = slice(xs.start + w.off_x, xs.stop + w.off_x)
            for q in range(19):
                dev = self._fetch_planes(self.dist_addrs[cur] + q * self.stride * self.isz, w.z, 1)[0][ys, xs]
                ref = w.dist[cur][q, w.t][wys, wxs]
                ok = np.isfinite(ref)
                res['compared_values'] += int(ok.sum())
                if not np.array_equal(dev[ok], ref[ok]):
                    res['dist_exact'] = False
                    with np.errstate(invalid='ignore'):
                        res['dist_err'] = max(res['dist_err'], float(np.nanmax(np.abs(dev[ok] - ref[ok]))))
            res['nodes'] += (d.lat_ny - 2) * (d.lat_nx - 2)
            if fields and self.field_addrs is not None:
                ref_rho = w.rho[w.t][wys, wxs]
                wet = np.isfinite(ref_rho)
                dev_rho = self._fetch_planes(self.field_addrs[0], w.z, 1)[0][ys, xs]
                if wet.any():
                    res['rho_err'] = max(res['rho_err'], float(np.max(np.abs(dev_rho[wet] - ref_rho[wet]) / np.abs(ref_rho[wet]))))
                    for c in range(3):
                        dev_v = self._fetch_planes(self.field_addrs[1 + c], w.z, 1)[0][ys, xs]
                        res['v_abs_err'] = max(res['v_abs_err'], float(np.max(np.abs(dev_v[wet] - w.v[c][w.t][wys, wxs][wet]))))
        return res


class SeamCheck(PlaneCheck):
    """Plane windows of ONE SLAB of a box that is cut into slabs along `axis` (0 = x, 1 = y, 2 = z) and exchanges its
    face layers with ring neighbours every step: the windows reach E = RADIUS + 1 layers beyond the slab along the split
    axis and hold there what the NEIGHBOURS' arrays hold -- a window is then a piece of the global box that straddles the
    seam, the oracle advances it with no notion of slabs, and the sampled planes are compared over the slab's full
    extent: the seam layers (x = 1, x = nx columns of an x-slab; planes z = 1, z = nz of a z-slab) are checked bit for
    bit against what a transfer that failed or arrived late would have spoilt.

    swap(low, high) -> (from_down, from_up): `low` / `high` are this slab's first / last E real layers along the split
    axis (numpy, [copy][19, ...]); the call returns the last E layers of the down neighbour and the first E layers of
    the up neighbour (a ring of one: (high, low)).  The slab itself is a fluid-only periodic box (bench.py's slabs):
    the unsplit axes are wrapped in-sweep.  Seed with the device idle and the x-face buffers materialised."""

    def __init__(self, backend, desc, zs, dist_addrs, stride, field_addrs, axis, swap):
        PlaneCheck.__init__(self, backend, desc, None, [], dist_addrs, stride, field_addrs)
        self.axis, self.swap, self.E = int(axis), swap, RADIUS + 1
        E = self.E
        n = [desc.lat_nx - 2, desc.lat_ny - 2, desc.lat_nz - 2]
        if n[self.axis] < E:
            raise ValueError('slab thinner than %d layers along the split axis' % E)
        for z in zs:
            if self.axis == 2:
                w = PlaneWindow.__new__(PlaneWindow)
                w.z, w.t = z, E
                w.planes = [z - E + k for k in range(W)]        # local plane indices, may leave [1, nz]
                wd = _clone_desc(desc, lat_nz=W, arr_nz=W, dist_stride=0)
                wd.periodic_fused[2] = wd.periodic_local[2] = 0
                w.off_y = w.off_x = 0
            else:
                w = PlaneWindow(desc, None, z)
                ext = 2 * (E - 1)
                if self.axis == 0:
                    lat = desc.lat_nx + ext
                    wd = _clone_desc(w.desc, lat_nx=lat, arr_nx=(lat + 31) // 32 * 32)
                    w.off_y, w.off_x = 0, E - 1
                else:
                    lat = desc.lat_ny + ext
                    wd = _clone_desc(w.desc, lat_ny=lat, arr_ny=lat)
                    w.off_y, w.off_x = E - 1, 0
                wd.periodic_fused[self.axis] = wd.periodic_local[self.axis] = 0
            w.desc, w.full = wd, desc
            w.o = OracleSim(wd)
            w.map, w.dist = None, None
            w.aa = desc.access_pattern == hipabi.SLF_AA
            w.pbc_axes = []
            self.windows.append(w)

    def _edge(self, block, lo):
        """The first (lo) / last E real layers of a [19, W, y, x] block along the split axis."""
        n = [self.desc.lat_nx - 2, self.desc.lat_ny - 2][self.axis]
        sl = slice(1, 1 + self.E) if lo else slice(n - self.E + 1, n + 1)
        return np.ascontiguousarray(block[:, :, :, sl] if self.axis == 0 else block[:, :, sl, :])

    def seed(self, iteration):
        self.iteration = int(iteration)
        d, E = self.desc, self.E
        if self.axis == 2:
            nz = d.lat_nz - 2
            mine = [(np.stack([self._fetch_planes(a + q * self.stride * self.isz, 1, E) for q in range(19)]),
                     np.stack([self._fetch_planes(a + q * self.stride * self.isz, nz - E + 1, E) for q in range(19)]))
                    for a in self.dist_addrs]
            down, up = self.swap([m[0] for m in mine], [m[1] for m in mine])
            for w in self.windows:
                w.dist = []
                for c, a in enumerate(self.dist_addrs):
                    blk = np.empty((19, W, d.arr_ny, d.arr_nx), dtype=self.dtype)
                    for k, p in enumerate(w.planes):
                        if p < 1:
                            blk[:, k] = down[c][:, p + E - 1]
                        elif p > nz:
                            blk[:, k] = up[c][:, p - nz - 1]
                        else:
                            for q in range(19):
                                blk[q, k] = self._fetch_planes(a + q * self.stride * self.isz, p, 1)[0]
                    w.dist.append(blk)
        else:
            local = [[self._fetch_window(w, a) for a in self.dist_addrs] for w in self.windows]
            down, up = self.swap([[self._edge(b, True) for b in per_w] for per_w in local],
                                 [[self._edge(b, False) for b in per_w] for per_w in local])
            n = [d.lat_nx - 2, d.lat_ny - 2][self.axis]
            for i, w in enumerate(self.windows):
                w.dist = []
                for c in range(len(self.dist_addrs)):
                    blk = np.zeros((19, W, w.desc.arr_ny, w.desc.arr_nx), dtype=self.dtype)
                    src = local[i][c]
                    if self.axis == 0:
                        blk[:, :, :d.arr_ny, E:E + n] = src[:, :, :, 1:n + 1]
                        blk[:, :, :d.arr_ny, 0:E] = down[i][c]
                        blk[:, :, :d.arr_ny, E + n:E + n + E] = up[i][c]
                    else:
                        blk[:, :, E:E + n, :] = src[:, :, 1:n + 1, :]
                        blk[:, :, 0:E, :] = down[i][c]
                        blk[:, :, E + n:E + n + E, :] = up[i][c]
                    w.dist.append(blk)
        for w in self.windows:
            w.rho = np.full(w.o.shape, np.inf, dtype=self.dtype)
            w.v = [np.full(w.o.shape, np.inf, dtype=self.dtype) for _ in range(3)]


def chunk_boundary_planes(placed, desc, stride, limit=4):
    """z-planes that contain a boundary between two physical chunks of a placed distribution array
    (sailfish_amd/placement.py): the rows there are where a wrong chunk mapping would show."""
    isz = 4 if desc.precision == 4 else 8
    nxy = desc.arr_nx * desc.arr_ny
    nz = desc.lat_nz - 2
    out = []
    for pb in placed:
        k = 1
        while k * pb.part_bytes < pb.total and len(out) < limit:
            off = k * pb.part_bytes - (pb.addr - pb.va)      # byte offset from the array's first element
            elem = off // isz
            z = int((elem % stride) // nxy)
            if RADIUS + 1 <= z <= nz - RADIUS - 1 and z not in out:
                out.append(z)
            k += max(1, (pb.total // pb.part_bytes) // limit)
    return out


class GlobalCheck(object):
    """Plane windows of the UNDIVIDED box of a run that is cut into `world` slabs along `axis`, one slab per rank:
    every rank hands in its share of the seven planes around each sampled GLOBAL plane (gather(obj) -> list of every
    rank's obj on rank 0, None elsewhere -- bench.py passes torch.distributed.gather_object), rank 0 merges them into a
    window of the undivided periodic box (full x / y extent of the GLOBAL box, every axis wrapped in-sweep: the oracle
    has no notion of slabs, ranks or halos), advances it two steps and compares the sampled plane with the merged
    plane of the slabs' arrays after their two steps -- every population of every real node, bit for bit.  Where
    SeamCheck looks at one slab and three layers of its neighbours, this looks at whole planes across ALL the seams at
    once (config 4: eight subdomains, seven inner seams and the wrap 7 -> 0).  The slabs are fluid-only periodic boxes
    (bench.py's); seed with the devices idle and the x-face buffers materialised."""

    def __init__(self, backend, desc, zs, dist_addrs, stride, axis, rank, world, gather):
        self.pc = PlaneCheck(backend, desc, None, [], dist_addrs, stride, None)
        self.desc, self.axis, self.rank, self.world, self.gather = desc, int(axis), int(rank), int(world), gather
        self.n = [desc.lat_nx - 2, desc.lat_ny - 2, desc.lat_nz - 2]
        g = list(self.n)
        g[self.axis] *= self.world
        self.g = g
        gd = _clone_desc(desc, lat_nx=g[0] + 2, lat_ny=g[1] + 2, lat_nz=g[2] + 2, arr_nx=(g[0] + 2 + 31) // 32 * 32,
                         arr_ny=g[1] + 2, arr_nz=g[2] + 2, dist_stride=0)
        for a in range(3):
            gd.periodic_fused[a] = gd.periodic_local[a] = 1
        self.gdesc = gd
        self.windows = [PlaneWindow(gd, None, z) for z in zs]       # planes: GLOBAL indices 1 .. g[2], wrapped
        self.dtype = self.pc.dtype
        self.iteration = None
        self.aa = desc.access_pattern == hipabi.SLF_AA

    # -- this rank's share of a list of global planes --------------------------------------------------------------
    def _share(self, planes, addr):
        """[19, len(planes), ny, nx] real nodes of the planes this rank holds (z split: None for the planes of other
        ranks), as a dict {position in `planes`: [19, ny, nx]}."""
        pc, n = self.pc, self.n
        out = {}
        for k, p in enumerate(planes):
            if self.axis == 2:
                if (p - 1) // n[2] != self.rank:
                    continue
                lp = (p - 1) % n[2] + 1
            else:
                lp = p
            blk = np.empty((19, n[1], n[0]), dtype=self.dtype)
            for q in range(19):
                blk[q] = pc._fetch_planes(addr + q * pc.stride * pc.isz, lp, 1)[0][1:n[1] + 1, 1:n[0] + 1]
            out[k] = blk
        return out

    def _merge(self, shares, count):
        """Rank 0: the shares of every rank -> [19, count, g_ny + 2, g_arr_nx] (ghost rows / columns never read: zero)."""
        gd, n = self.gdesc, self.n
        blk = np.zeros((19, count, gd.arr_ny, gd.arr_nx), dtype=self.dtype)
        filled = np.zeros(count, dtype=np.int64)
        for r, sh in enumerate(shares):
            oy = 1 + (r * n[1] if self.axis == 1 else 0)
            ox = 1 + (r * n[0] if self.axis == 0 else 0)
            for k, part in sh.items():
                blk[:, k, oy:oy + n[1], ox:ox + n[0]] = part
                filled[k] += 1
        want = 1 if self.axis == 2 else self.world
        if not np.all(filled == want):
            raise RuntimeError('global window: planes covered %s times, expected %d' % (filled.tolist(), want))
        return blk

    def _current(self, iteration):
        return 0 if len(self.pc.dist_addrs) == 1 else (iteration & 1)

    def seed(self, iteration):
        self.iteration = int(iteration)
        cur = self._current(self.iteration)
        for w in self.windows:
            shares = self.gather(self._share(w.planes, self.pc.dist_addrs[cur]))
            if self.rank == 0:
                blk = self._merge(shares, W)
                w.dist = [blk] if len(self.pc.dist_addrs) == 1 else [None, None]
                if len(w.dist) == 2:
                    w.dist[cur], w.dist[1 - cur] = blk, np.zeros_like(blk)
                w.rho = np.full(w.o.shape, np.inf, dtype=self.dtype)
                w.v = [np.full(w.o.shape, np.inf, dtype=self.dtype) for _ in range(3)]

    def advance(self, steps=2):
        if self.rank == 0:
            self.pc.windows, self.pc.iteration = self.windows, self.iteration
            self.pc.advance(steps, save_last=False)
            self.pc.windows = []
        self.iteration += steps

    def compare(self):
        """Rank 0: {'planes', 'compared_values', 'dist_exact', 'dist_err'}; other ranks: None (they only hand in)."""
        cur = self._current(self.iteration)
        res = {'planes': [w.z for w in self.windows], 'compared_values': 0, 'dist_exact': True, 'dist_err': 0.0,
               'box': 'x'.join(str(v) for v in self.g), 'slabs': self.world}
        for w in self.windows:
            shares = self.gather(self._share([w.z], self.pc.dist_addrs[cur]))
            if self.rank != 0:
                continue
            dev = self._merge(shares, 1)[:, 0, 1:self.g[1] + 1, 1:self.g[0] + 1]
            ref = w.dist[cur][:, w.t, 1:self.g[1] + 1, 1:self.g[0] + 1]
            res['compared_values'] += int(ref.size)
            if not np.array_equal(dev, ref):
                res['dist_exact'] = False
                with np.errstate(invalid='ignore'):
                    res['dist_err'] = max(res['dist_err'], float(np.nanmax(np.abs(dev - ref))))
        return res if self.rank == 0 else None
