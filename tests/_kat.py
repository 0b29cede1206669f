"""Builds simulations from the propagation known-answer tables captured from the reference's
regtest/subdomains/{2d,3d}_propagation.py (tests/golden/propagation_kat.json)."""
import json
import os

import numpy as np

from sailfish_amd import sym
from sailfish_amd.geo import LBGeometry2D, LBGeometry3D
from sailfish_amd.lb_single import LBFluidSim
from sailfish_amd.subdomain import Subdomain2D, Subdomain3D, SubdomainSpec2D, SubdomainSpec3D

with open(os.path.join(os.path.dirname(__file__), 'golden', 'propagation_kat.json')) as fh:
    KATS = json.load(fh)


def all_runs():
    """[(id string, run dict)] for every controller run of every reference test method."""
    out = []
    for fname, tests in sorted(KATS.items()):
        for tname, runs in sorted(tests.items()):
            for k, run in enumerate(runs):
                out.append(('%s:%s:%d' % (fname[:2], tname, k), run))
    return out


def supported(run):
    return run['grid'] in ('D2Q9', 'D3Q19')


class Blank2D(Subdomain2D):
    def boundary_conditions(self, hx, hy):
        pass

    def initial_conditions(self, sim, hx, hy):
        pass


class Blank3D(Subdomain3D):
    def boundary_conditions(self, hx, hy, hz):
        pass

    def initial_conditions(self, sim, hx, hy, hz):
        pass


def make_classes(run):
    dim = run['dim']
    grid = sym.lookup_grid(run['grid'])

    class KatGeo(LBGeometry2D if dim == 2 else LBGeometry3D):
        def subdomains(self, n=None):
            cls = SubdomainSpec2D if dim == 2 else SubdomainSpec3D
            return [cls(tuple(s['location']), tuple(s['size'])) for s in run['subdomains']]

    class KatSim(LBFluidSim):
        subdomain = Blank2D if dim == 2 else Blank3D

        def initial_conditions(self, runner):
            """Same as the reference tests: zero everything, write the tagged values into both copies."""
            dbuf = runner._debug_get_dist()
            dbuf[:] = 0.0
            for inp in run['inputs']:
                if inp['subdomain'] == runner._spec.id:
                    q = grid.vec_idx(inp['vec'])
                    dbuf[(q,) + tuple(inp['pos'])] = inp['value']
            runner._debug_set_dist(dbuf, copy=0)
            if runner.config.access_pattern == 'AB':
                runner._debug_set_dist(dbuf, copy=1)

    cfg = dict(lat_nx=run['lat'][0], lat_ny=run['lat'][1], grid=run['grid'], access_pattern=run['access_pattern'],
               periodic_x=run['periodic'][0], periodic_y=run['periodic'][1], mem_alignment=run['mem_alignment'],
               relaxation_enabled=run['relaxation_enabled'], max_iters=run['max_iters'], visc=1.0)
    if dim == 3:
        cfg.update(lat_nz=run['lat'][2], periodic_z=run['periodic'][2])
    return KatSim, KatGeo, cfg, grid


def check(run, grid, get_dist):
    """get_dist(subdomain id, iteration) -> [Q, (nz,) ny, arr_nx] array."""
    bad = []
    for e in run['expects']:
        d = get_dist(e['subdomain'], e['iteration'])
        got = d[(grid.vec_idx(e['vec']),) + tuple(e['pos'])]
        if got != np.float32(e['value']):
            bad.append((e, float(got)))
    assert not bad, '%d of %d expected slots differ, first: %r' % (len(bad), len(run['expects']), bad[:3])
