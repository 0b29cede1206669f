"""Which populations travel between which subdomains (the halo contract).

The reference describes every subdomain-to-subdomain link with slice algebra
(sailfish/subdomain_connection.py:238-534: src_slice, dst_low, dst_slice,
dst_full_buf_slice, dst_partial_map, ...), packs contiguous faces with
Collect/DistributeContinuousData and strided faces / edge nodes with index
lists (CollectSparseData).  Here everything is an index list, derived from one
rule -- *route by the owner of the global node position*:

  push steps (AB, odd AA step; reference subdomain_runner.py:1064-1139, E in SURVEY §3):
    a population pushed from a real node of A into A's ghost node g belongs to
    the real node of the subdomain B that owns g's (periodically wrapped) global
    position, same slot.
  even AA step (in-place, opposite slots; reference Appendix A.5 / 3d_propagation.py:706-720):
    the real node of A at x + e_q that will *pull* f_q next step from ghost g
    needs slot opp(q) of the real node that owns g's global position.

Both sides enumerate the same (node, population) pairs and sort them by
(population, global z, y, x) of the real node involved, so a message needs no
header.  Faces, edges and corners, periodic images and mixed "locally periodic
along y, decomposed along x" layouts all fall out of the same rule.
Populations whose owner is the subdomain itself are the job of the local
periodic-boundary kernels.
"""
import numpy as np



class HaloLink(object):
    """Index lists (uint64, q * dist_stride + node index) of one directed neighbour relation."""

    def __init__(self, neighbour_id):
        self.neighbour_id = neighbour_id
        self.push_send = self.push_recv = self.pull_send = self.pull_recv = None

    @property
    def send_count(self):
        return len(self.push_send)

    @property
    def recv_count(self):
        return len(self.push_recv)


def _owner_of(points, specs, gsize, periodic):
    """points: (N, dim) global positions (may lie outside the box).  Returns (owner ids, wrapped points);
    owner -1 = outside a non-periodic boundary."""
    pts = points.copy()
    valid = np.ones(len(pts), dtype=bool)
    for a in range(pts.shape[1]):
        if periodic[a]:
            pts[:, a] %= gsize[a]
        else:
            valid &= (pts[:, a] >= 0) & (pts[:, a] < gsize[a])
    owner = np.full(len(pts), -1, dtype=np.int64)
    for s in specs:
        m = valid.copy()
        for a in range(pts.shape[1]):
            m &= (pts[:, a] >= s.location[a]) & (pts[:, a] < s.location[a] + s.size[a])
        owner[m] = s.id
    return owner, pts


def _shell_coords(n):
    """Local coordinates (x, y[, z]) of the ghost-including boundary region: every node of the
    (n+2)^dim box that is a ghost or lies in the first / last real layer."""
    lat = [k + 2 for k in n]
    axes = [np.arange(k) for k in lat]
    grids = np.meshgrid(*axes, indexing='ij')
    coords = np.stack([g.ravel() for g in grids], axis=1)
    near = np.zeros(len(coords), dtype=bool)
    for a in range(len(n)):
        near |= (coords[:, a] <= 1) | (coords[:, a] >= lat[a] - 2)
    return coords[near]


def _shell_coords_fast(n):
    """Same as _shell_coords without materialising the full box (large subdomains)."""
    dim = len(n)
    lat = [k + 2 for k in n]
    pieces = []
    for a in range(dim):
        for layer in (0, 1, lat[a] - 2, lat[a] - 1):
            if layer < 0 or layer >= lat[a]:
                continue
            axes = []
            for b in range(dim):
                if b == a:
                    axes.append(np.array([layer]))
                elif b < a:
                    # layers of earlier axes were already emitted completely
                    axes.append(np.arange(2, lat[b] - 2))
                else:
                    axes.append(np.arange(lat[b]))
            if any(len(x) == 0 for x in axes):
                continue
            grids = np.meshgrid(*axes, indexing='ij')
            pieces.append(np.stack([g.ravel() for g in grids], axis=1))
    coords = np.concatenate(pieces, axis=0)
    return np.unique(coords, axis=0)


def build_halo_links(spec, specs, gsize, periodic, grid, arr_shape, dist_stride, fused=None):
    """Returns {neighbour_id: HaloLink} for subdomain `spec`.

    arr_shape = (arr_nx, arr_ny[, arr_nz]) padded in-memory size of `spec`;
    gsize / periodic: global box and its periodicity, x first;
    fused[a]: axis a is wrapped inside the sweep of this subdomain (no ghost rows are used along it,
    neighbour arithmetic is modulo the subdomain size there).
    """
    dim = grid.dim
    n = list(spec.size)
    lat = [k + 2 for k in n]
    origin = np.array(spec.location, dtype=np.int64)
    coords = _shell_coords_fast(n).astype(np.int64)
    fused = [bool(x) for x in (fused or [0] * dim)][:dim]
    is_real = np.all((coords >= 1) & (coords <= np.array(n)), axis=1)
    ghosts = coords[~is_real]
    for a in range(dim):
        if fused[a]:
            ghosts = ghosts[(ghosts[:, a] >= 1) & (ghosts[:, a] <= n[a])]
    reals = coords[is_real]
    stride = np.array([1, arr_shape[0], arr_shape[0] * arr_shape[1]][:dim], dtype=np.int64)

    def wrapf(c):
        c = c.copy()
        for a in range(dim):
            if fused[a]:
                c[:, a] = (c[:, a] - 1) % n[a] + 1
        return c

    def lin(c):
        return (c * stride).sum(axis=1).astype(np.uint64)

    def real_mask(c):
        return np.all((c >= 1) & (c <= np.array(n)), axis=1)

    g_owner, g_pos = _owner_of(ghosts + origin - 1, specs, gsize, periodic)
    basis = np.array(grid.basis, dtype=np.int64)
    opp = grid.idx_opposite
    ds = np.uint64(dist_stride)

    acc = {}

    def add(kind, nid, q_slot, q_key, lin_idx, gpos):
        if len(lin_idx) == 0:
            return
        d = acc.setdefault(int(nid), {}).setdefault(kind, [])
        key = np.concatenate([np.full((len(lin_idx), 1), q_key, dtype=np.int64), gpos[:, ::-1]], axis=1)
        d.append((key, np.uint64(q_slot) * ds + lin_idx))

    for q in range(1, grid.Q):
        e = basis[q]
        # ---- ghosts of this subdomain
        src_real = real_mask(wrapf(ghosts - e))   # pushed from one of my real nodes
        puller_real = real_mask(wrapf(ghosts + e))  # one of my real nodes pulls f_q from here
        for nid in np.unique(g_owner):
            if nid < 0 or nid == spec.id:
                continue
            m = (g_owner == nid) & src_real
            add('push_send', nid, q, q, lin(ghosts[m]), g_pos[m])
            m = (g_owner == nid) & puller_real
            add('pull_recv', nid, opp[q], q, lin(ghosts[m]), g_pos[m])
        # ---- real boundary nodes of this subdomain
        r_pos = reals + origin - 1
        src_owner, _ = _owner_of(r_pos - e, specs, gsize, periodic)     # who pushed f_q into this node
        src_is_ghost = ~real_mask(wrapf(reals - e))
        pul_owner, _ = _owner_of(r_pos + e, specs, gsize, periodic)     # who pulls f_q from this node
        pul_is_ghost = ~real_mask(wrapf(reals + e))
        for nid in np.unique(np.concatenate([src_owner, pul_owner])):
            if nid < 0 or nid == spec.id:
                continue
            m = (src_owner == nid) & src_is_ghost
            add('push_recv', nid, q, q, lin(reals[m]), r_pos[m])
            m = (pul_owner == nid) & pul_is_ghost
            add('pull_send', nid, opp[q], q, lin(reals[m]), r_pos[m])

    links = {}
    for nid, kinds in acc.items():
        link = HaloLink(nid)
        for kind in ('push_send', 'push_recv', 'pull_send', 'pull_recv'):
            parts = kinds.get(kind, [])
            if parts:
                keys = np.concatenate([p[0] for p in parts], axis=0)
                idx = np.concatenate([p[1] for p in parts])
                order = np.lexsort(keys.T[::-1])
                setattr(link, kind, np.ascontiguousarray(idx[order], dtype=np.uint64))
            else:
                setattr(link, kind, np.zeros(0, dtype=np.uint64))
        links[nid] = link
    return links


def connect_subdomains(specs, gsize, periodic):
    """Face adjacency (incl. periodic images) and local periodicity of a set of SubdomainSpecs --
    the effect of the reference's LBGeometryProcessor (controller.py:130-269) on has_face_conn(),
    connecting_subdomains() and enable_local_periodicity()."""
    dim = specs[0].dim
    for i, s in enumerate(specs):
        if s.id is None:
            s.id = i
        s._clear_connections()
    for s in specs:
        for a in range(dim):
            if periodic[a] and s.location[a] == 0 and s.location[a] + s.size[a] == gsize[a]:
                s.enable_local_periodicity(a)
    for s in specs:
        for a in range(dim):
            for d in (-1, 1):
                # a probe layer just outside face (a, d)
                lo = list(s.location)
                hi = [o + k for o, k in zip(s.location, s.size)]
                pos = s.location[a] - 1 if d < 0 else s.location[a] + s.size[a]
                if pos < 0 or pos >= gsize[a]:
                    if not periodic[a]:
                        continue
                    pos %= gsize[a]
                for t in specs:
                    if t.id == s.id and s._periodicity[a]:
                        continue
                    if not (t.location[a] <= pos < t.location[a] + t.size[a]):
                        continue
                    overlap = True
                    for b in range(dim):
                        if b == a:
                            continue
                        if min(hi[b], t.location[b] + t.size[b]) <= max(lo[b], t.location[b]):
                            overlap = False
                    if overlap and t.id != s.id:
                        s._add_connection(s.axis_dir_to_face(a, d), t.id)
    return specs


def ghost_owned_by_others(spec, specs, gsize, periodic):
    """Boolean array (numpy axis order, ghosts included): ghost nodes whose global position is a real
    node of another subdomain."""
    dim = spec.dim
    lat = [k + 2 for k in spec.size]
    out = np.zeros(list(reversed(lat)), dtype=bool)
    coords = _shell_coords_fast(list(spec.size)).astype(np.int64)
    is_real = np.all((coords >= 1) & (coords <= np.array(spec.size)), axis=1)
    ghosts = coords[~is_real]
    owner, _ = _owner_of(ghosts + np.array(spec.location) - 1, specs, gsize, periodic)
    sel = ghosts[(owner >= 0) & (owner != spec.id)]
    out[tuple(sel[:, a] for a in reversed(range(dim)))] = True
    return out


class MacroLink(object):
    """Node index lists (uint64) of one neighbour relation for macroscopic fields."""

    def __init__(self, neighbour_id):
        self.neighbour_id = neighbour_id
        self.send = np.zeros(0, dtype=np.uint64)
        self.recv = np.zeros(0, dtype=np.uint64)


def _foreign_ghosts(spec, specs, gsize, periodic, fused):
    """Ghost nodes of `spec` whose global position is a real node of another subdomain:
    (local coords, owner ids, wrapped global positions), in a canonical order (local z, y, x)."""
    dim = spec.dim
    n = list(spec.size)
    coords = _shell_coords_fast(n).astype(np.int64)
    is_real = np.all((coords >= 1) & (coords <= np.array(n)), axis=1)
    ghosts = coords[~is_real]
    for a in range(dim):
        if fused[a]:
            ghosts = ghosts[(ghosts[:, a] >= 1) & (ghosts[:, a] <= n[a])]
    owner, pos = _owner_of(ghosts + np.array(spec.location, dtype=np.int64) - 1, specs, gsize, periodic)
    keep = (owner >= 0) & (owner != spec.id)
    ghosts, owner, pos = ghosts[keep], owner[keep], pos[keep]
    order = np.lexsort(ghosts.T)
    return ghosts[order], owner[order], pos[order]


def build_macro_links(spec, specs, gsize, periodic, arr_shape, fused_of):
    """{neighbour_id: MacroLink} for the exchange of macroscopic fields read by non-local models
    (Shan-Chen: psi(rho) of the 18 neighbours; reference _send_macro/_recv_macro, subdomain_runner.py:
    2033-2100): every ghost node of a subdomain receives the value of the real node that owns its global
    position.  Message order = canonical order of the *receiver's* ghost nodes; the sender derives the
    same list from the receiver's spec.  fused_of(s) -> per-axis flags "s wraps this axis in its kernels"."""
    dim = spec.dim
    stride = np.array([1, arr_shape[0], arr_shape[0] * arr_shape[1]][:dim], dtype=np.int64)
    origin = np.array(spec.location, dtype=np.int64)

    def lin(c):
        return np.ascontiguousarray((c * stride).sum(axis=1), dtype=np.uint64)

    links = {}
    g, owner, _ = _foreign_ghosts(spec, specs, gsize, periodic, fused_of(spec))
    for nid in np.unique(owner):
        links.setdefault(int(nid), MacroLink(int(nid))).recv = lin(g[owner == nid])
    for t in specs:
        if t.id == spec.id:
            continue
        g, owner, pos = _foreign_ghosts(t, specs, gsize, periodic, fused_of(t))
        m = owner == spec.id
        if m.any():
            links.setdefault(int(t.id), MacroLink(int(t.id))).send = lin(pos[m] - origin + 1)
    return links
