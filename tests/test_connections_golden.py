"""a16: which populations travel between which subdomains.  The reference describes every directed
connection with slice algebra (LBConnection.make, subdomain_connection.py:399-440: `dists`, the full region
`dst_slice`, the partial nodes `dst_partial_map`, extra connection objects for periodic images that touch a face at an
edge or a corner); sailfish_amd/subdomain_connection.py derives index lists from ONE rule -- route by the owner of the
global node position.  tests/golden/connections.json holds the reference's objects expanded into plain sets
{(population, position in the receiver's real-node coordinates)} per ordered pair of subdomains
(tools/capture_connections.py, run where /root/reference exists): the index lists must name exactly those."""
import json
import os

import numpy as np
import pytest

from sailfish_amd import subdomain_connection, sym
from sailfish_amd.subdomain import SubdomainSpec2D, SubdomainSpec3D

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'connections.json')))


@pytest.mark.parametrize('name', sorted(GOLD))
def test_halo_index_lists_name_what_the_reference_connections_transfer(name):
    case = GOLD[name]
    dim = len(case['gsize'])
    grid = sym.D2Q9 if dim == 2 else sym.D3Q19
    cls = SubdomainSpec2D if dim == 2 else SubdomainSpec3D
    specs = [cls(tuple(loc), tuple(size)) for loc, size in case['boxes']]
    for i, s in enumerate(specs):
        s.id = i
        s.set_actual_size(1)
    subdomain_connection.connect_subdomains(specs, case['gsize'], case['periodic'])
    got = {}
    for recv in specs:
        arr = [n + 2 for n in recv.size]                       # no x padding: indices decode with the lattice size itself
        stride = int(np.prod(arr))
        links = subdomain_connection.build_halo_links(recv, specs, case['gsize'], case['periodic'], grid, arr, stride)
        for sender_id, link in links.items():
            idx = link.push_recv.astype(np.int64)
            q, node = idx // stride, idx % stride
            coords = []
            for a in range(dim):
                coords.append(node % arr[a] - 1)               # real-node coordinates (ghost envelope of 1 removed)
                node = node // arr[a]
            items = set(zip(*([q.tolist()] + [c.tolist() for c in coords])))
            if items:
                got['%d->%d' % (sender_id, recv.id)] = items
            # both ends of a link list the same pairs in the same order: a message needs no header
            back = subdomain_connection.build_halo_links(specs[sender_id], specs, case['gsize'], case['periodic'], grid,
                                                         [n + 2 for n in specs[sender_id].size],
                                                         int(np.prod([n + 2 for n in specs[sender_id].size])))
            assert len(back[recv.id].push_send) == len(link.push_recv)
    want = dict((k, set(tuple(t) for t in v)) for k, v in case['pairs'].items())
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], (name, k, sorted(got[k] ^ want[k])[:6])
