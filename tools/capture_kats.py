#!/usr/bin/env python3
"""Propagation known-answer tables from the reference's regression tests
(regtest/subdomains/2d_propagation.py, 3d_propagation.py), captured *mechanically*:

the reference's test modules are executed unchanged, with three seams replaced by
recorders -- `LBSimulationController` (records the configuration, the subdomain layout and
the values the test writes with runner._debug_set_dist), `numpy.load` of a distribution
dump (returns a proxy that records which [dist, (z,) y, x] slot of which subdomain /
iteration the test reads) and `numpy.testing.assert_equal` (records the expected value).
Nothing is computed.  Output: tests/golden/propagation_kat.json =
pure data: inputs and expected outputs of every test method.

    python tools/capture_kats.py
"""
import importlib.util
import json
import os
import re
import sys
import unittest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: F401
import types

import numpy as np

# regtest.subdomains.util imports utils.merge_subdomains (needs nothing we use) -> stub it
_u = types.ModuleType('utils')
_um = types.ModuleType('utils.merge_subdomains')
_um.merge_subdomains = lambda *a, **k: None
sys.modules['utils'] = _u
sys.modules['utils.merge_subdomains'] = _um

from sailfish import sym  # noqa: E402
from sailfish.config import LBConfig  # noqa: E402
from sailfish.geo import LBGeometry2D, LBGeometry3D  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'propagation_kat.json')
CURRENT = {'runs': []}


class _Logger(object):
    def debug(self, *a):
        pass
    info = warning = error = debug


class FakeSpecRunner(object):
    def __init__(self, spec, grid, mem_alignment, run):
        self._spec = spec
        self._grid = grid
        self._run = run
        self._recorded = False
        lat = [n + 2 for n in spec.size]
        arr_nx = ((lat[0] + mem_alignment - 1) // mem_alignment) * mem_alignment
        self._shape = [grid.Q] + list(reversed(lat[1:])) + [arr_nx]

    def _debug_get_dist(self, *a, **k):
        return np.zeros(self._shape, dtype=np.float32)

    def _debug_set_dist(self, dbuf, output=True, *a, **k):
        if self._recorded:
            return
        self._recorded = True
        for idx in np.argwhere(dbuf != 0):
            basis = [int(c) for c in self._grid.basis[int(idx[0])]]
            self._run['inputs'].append({'subdomain': int(self._spec.id), 'vec': basis,
                                        'pos': [int(i) for i in idx[1:]], 'value': float(dbuf[tuple(idx)])})


class FakeController(object):
    def __init__(self, lb_class, lb_geo=None, default_config=None):
        self.lb_class, self.lb_geo, self.default_config = lb_class, lb_geo, default_config

    def run(self, ignore_cmdline=False):
        dim = self.lb_class.subdomain.dim
        d = {'periodic_x': False, 'periodic_y': False, 'periodic_z': False, 'access_pattern': 'AB',
             'subdomains': 1, 'conn_axis': 'x', 'grid': 'D2Q9' if dim == 2 else 'D3Q19', 'mem_alignment': 32,
             'block_size': 64, 'precision': 'single', 'node_addressing': 'direct', 'max_iters': 0,
             'lat_nz': 1, 'visc': 1.0, 'model': 'bgk'}
        self.lb_class.update_defaults(d)
        if self.default_config:
            d.update(self.default_config)
        cfg = LBConfig()
        for k, v in d.items():
            setattr(cfg, k, v)
        cfg.logger = _Logger()
        cfg.relaxation_enabled = True
        cfg.propagation_enabled = True
        cfg.time_dependence = cfg.space_dependence = False
        cfg.use_link_tags = True
        cfg.incompressible = False
        cfg.minimize_roundoff = False
        self.lb_class.modify_config(cfg)
        geo_cls = self.lb_geo or (LBGeometry2D if dim == 2 else LBGeometry3D)
        specs = geo_cls(cfg).subdomains()
        grid = [g for g in sym.KNOWN_GRIDS if g.__name__ == cfg.grid][0]
        run = {'dim': dim, 'grid': cfg.grid, 'access_pattern': cfg.access_pattern,
               'lat': [cfg.lat_nx, cfg.lat_ny] + ([cfg.lat_nz] if dim == 3 else []),
               'periodic': [bool(cfg.periodic_x), bool(cfg.periodic_y)] + ([bool(cfg.periodic_z)] if dim == 3 else []),
               'relaxation_enabled': bool(cfg.relaxation_enabled), 'max_iters': int(cfg.max_iters),
               'mem_alignment': int(cfg.mem_alignment),
               'subdomains': [{'location': [int(x) for x in s.location], 'size': [int(x) for x in s.size]}
                              for s in specs],
               'inputs': [], 'expects': []}
        sim = self.lb_class(cfg)
        for i, s in enumerate(specs):
            s.id = i
            sim.initial_conditions(FakeSpecRunner(s, grid, cfg.mem_alignment, run))
        CURRENT['runs'].append(run)
        CURRENT['grid'] = grid


class Token(object):
    def __init__(self, sid, it, idx):
        self.sid, self.it, self.idx = sid, it, idx


class Recorder(object):
    def __init__(self, sid, it):
        self.sid, self.it = sid, it

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        return Token(self.sid, self.it, idx)


def fake_load(fname, *a, **k):
    m = re.search(r'_dists\.(\d+)\.(\d+)\.npz$', str(fname))
    assert m, fname
    return {'arr_0': Recorder(int(m.group(1)), int(m.group(2)))}


def fake_assert_equal(actual, desired, *a, **k):
    assert isinstance(actual, Token), 'unexpected assert_equal on %r' % (actual,)
    run = CURRENT['runs'][-1]
    grid = CURRENT['grid']
    idx = actual.idx
    assert all(isinstance(i, (int, np.integer)) for i in idx), idx
    run['expects'].append({'subdomain': actual.sid, 'iteration': actual.it,
                           'vec': [int(c) for c in grid.basis[int(idx[0])]],
                           'pos': [int(i) for i in idx[1:]], 'value': float(desired)})


def capture(path):
    name = 'kat_' + os.path.basename(path)[:-3]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.LBSimulationController = FakeController
    real_load, real_ae = np.load, np.testing.assert_equal
    np.load, np.testing.assert_equal = fake_load, fake_assert_equal
    out = {}
    try:
        if hasattr(mod, 'setUpModule'):
            mod.setUpModule()
        for cname in dir(mod):
            cls = getattr(mod, cname)
            if not (isinstance(cls, type) and issubclass(cls, unittest.TestCase)):
                continue
            for mname in sorted(dir(cls)):
                if not mname.startswith('test_'):
                    continue
                CURRENT['runs'] = []
                t = cls(mname)
                t.setUp()
                getattr(t, mname)()
                t.tearDown()
                out['%s.%s' % (cname, mname)] = CURRENT['runs']
        if hasattr(mod, 'tearDownModule'):
            mod.tearDownModule()
    finally:
        np.load, np.testing.assert_equal = real_load, real_ae
    return out


def main():
    res = {}
    for f in ('2d_propagation.py', '3d_propagation.py'):
        path = os.path.join('/root/reference/regtest/subdomains', f)
        res[f] = capture(path)
        n_exp = sum(len(r['expects']) for runs in res[f].values() for r in runs)
        print(f, len(res[f]), 'test methods,', n_exp, 'expected values')
    with open(OUT, 'w') as fh:
        json.dump(res, fh, indent=None, sort_keys=True)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
