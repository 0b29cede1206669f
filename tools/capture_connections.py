#!/usr/bin/env python3
"""Which populations travel between which subdomains, from the reference's own connection objects (authoring container
only).  For a set of decompositions -- 1-D slabs along every axis with and without periodic axes, 2 x 2 and 2 x 2 x 2
blocks, unequal slabs -- the reference's LBGeometryProcessor connects the subdomains (controller.py:130-269) and
SubdomainSpec.connect() / LBConnection.make() (subdomain.py, subdomain_connection.py:399-440) describe every directed
connection with slice algebra: the populations `dists`, the full region `dst_slice`, the partial nodes `dst_partial_map`.
This tool expands those objects into plain sets {(population, position in the RECEIVER's real-node coordinates)} per
ordered pair (sender, receiver) -- the facts behind the slice algebra -- and writes tests/golden/connections.json.
sailfish_amd/subdomain_connection.py derives its index lists from another rule (route by the owner of the global node
position); tests/test_connections_golden.py holds the two against each other.

    python tools/capture_connections.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: F401

import numpy as np

from sailfish import sym
from sailfish.controller import LBGeometryProcessor
from sailfish.subdomain import SubdomainSpec2D, SubdomainSpec3D

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'connections.json')


class _Cfg(object):
    def __init__(self, periodic):
        self.periodic_x, self.periodic_y = bool(periodic[0]), bool(periodic[1])
        self.periodic_z = bool(periodic[2]) if len(periodic) > 2 else False
        self.grid = 'D2Q9' if len(periodic) == 2 else 'D3Q19'


def slabs(gsize, n, axis):
    out, start = [], 0
    for i in range(n):
        ext = gsize[axis] // n + (gsize[axis] % n if i == n - 1 else 0)
        loc = [0] * len(gsize)
        size = list(gsize)
        loc[axis], size[axis] = start, ext
        out.append((tuple(loc), tuple(size)))
        start += ext
    return out


def blocks(gsize, cuts):
    import itertools
    edges = [[gsize[a] * k // cuts[a] for k in range(cuts[a] + 1)] for a in range(len(gsize))]
    out = []
    for idx in itertools.product(*[range(c) for c in cuts]):
        loc = tuple(edges[a][idx[a]] for a in range(len(gsize)))
        size = tuple(edges[a][idx[a] + 1] - edges[a][idx[a]] for a in range(len(gsize)))
        out.append((loc, size))
    return out


CASES = {
    '2d_x3_open': dict(gsize=(18, 8), periodic=(0, 0), boxes=slabs((18, 8), 3, 0)),
    '2d_y2_periodic_y': dict(gsize=(8, 12), periodic=(0, 1), boxes=slabs((8, 12), 2, 1)),
    '2d_x2_periodic_xy': dict(gsize=(12, 8), periodic=(1, 1), boxes=slabs((12, 8), 2, 0)),
    '2d_blocks_2x2_periodic_x': dict(gsize=(12, 10), periodic=(1, 0), boxes=blocks((12, 10), (2, 2))),
    '3d_x8_periodic_xyz': dict(gsize=(32, 6, 5), periodic=(1, 1, 1), boxes=slabs((32, 6, 5), 8, 0)),
    '3d_z2_open': dict(gsize=(8, 7, 10), periodic=(0, 0, 0), boxes=slabs((8, 7, 10), 2, 2)),
    '3d_y3_periodic_y': dict(gsize=(6, 13, 5), periodic=(0, 1, 0), boxes=slabs((6, 13, 5), 3, 1)),
    '3d_blocks_2x2x2_periodic_z': dict(gsize=(8, 8, 6), periodic=(0, 0, 1), boxes=blocks((8, 8, 6), (2, 2, 2))),
    '3d_blocks_2x2x1_periodic_xy': dict(gsize=(10, 8, 5), periodic=(1, 1, 0), boxes=blocks((10, 8, 5), (2, 2, 1))),
}


def expand(conn, dim, conn_axis, face_high, dst_size):
    """{(population, x, y[, z])} in the receiver's real-node coordinates (0-based) of one LBConnection."""
    slice_axes = [a for a in range(dim) if a != conn_axis]
    # the layer of the receiver that takes the data: populations leaving the sender through its HIGH face arrive in
    # the receiver's first real layer, through its LOW face in the last one
    layer = 0 if face_high else dst_size[conn_axis] - 1
    out = set()

    def pos(coords):
        p = [0] * dim
        p[conn_axis] = layer
        for a, c in zip(slice_axes, coords):
            p[a] = int(c)
        return tuple(p)
    if conn.dst_slice:
        ranges = [range(s.start, s.stop) for s in conn.dst_slice]
        grids = np.meshgrid(*ranges, indexing='ij')
        for coords in zip(*[g.ravel() for g in grids]):
            for q in conn.dists:
                out.add((int(q),) + pos(coords))
    for q, nodes in conn.dst_partial_map.items():
        for nd in np.atleast_2d(nodes):
            out.add((int(q),) + pos([int(c) + int(lo) for c, lo in zip(nd, conn.dst_low)]))
    return out


def main():
    res = {}
    for name, case in CASES.items():
        dim = len(case['gsize'])
        grid = sym.D2Q9 if dim == 2 else sym.D3Q19
        cls = SubdomainSpec2D if dim == 2 else SubdomainSpec3D
        specs = [cls(loc, size) for loc, size in case['boxes']]
        for s in specs:
            s.set_actual_size(1)
        specs = LBGeometryProcessor(specs, dim, case['gsize']).transform(_Cfg(case['periodic']))
        pairs = {}
        for s in specs:
            for face, nid in s.connecting_subdomains():
                conn_axis = s.face_to_axis(face)
                face_high = face % 2 == 1
                dst = specs[nid]
                # several connection objects may serve one (face, neighbour): the face itself and, across periodic
                # axes, the images that only touch it at an edge or a corner.  cpair.src: what s SENDS to nid.
                for cpair in s.get_connections(face, nid):
                    items = expand(cpair.src, dim, conn_axis, face_high, dst.size)
                    pairs.setdefault('%d->%d' % (s.id, nid), set()).update(items)
        # macroscopic fields (non-local models): ghost nodes of s that take the value of a real node of nid
        # (LBConnection.dst_macro_slice of s's own connection object, in s's ghost-including coordinates; the ghost
        # layer of `face`)
        macro = {}
        for s in specs:
            for face, nid in s.connecting_subdomains():
                conn_axis = s.face_to_axis(face)
                slice_axes = [a for a in range(dim) if a != conn_axis]
                layer = s.size[conn_axis] + 1 if face % 2 == 1 else 0
                for cpair in s.get_connections(face, nid):
                    ranges = [range(sl.start, sl.stop) for sl in cpair.src.dst_macro_slice]
                    grids = np.meshgrid(*ranges, indexing='ij')
                    for coords in zip(*[g.ravel() for g in grids]):
                        p = [0] * dim
                        p[conn_axis] = layer
                        for a, c in zip(slice_axes, coords):
                            p[a] = int(c)
                        macro.setdefault('%d->%d' % (nid, s.id), set()).add(tuple(p))
        res[name] = {'gsize': list(case['gsize']), 'periodic': [int(p) for p in case['periodic']],
                     'boxes': [[list(l), list(sz)] for l, sz in case['boxes']],
                     'pairs': dict((k, sorted(list(v))) for k, v in sorted(pairs.items())),
                     'macro_ghosts': dict((k, sorted(list(v))) for k, v in sorted(macro.items()))}
        print(name, dict((k, len(v)) for k, v in sorted(pairs.items())))
    with open(OUT, 'w') as fh:
        json.dump(res, fh, separators=(',', ':'))
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
