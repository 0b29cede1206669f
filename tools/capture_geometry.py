#!/usr/bin/env python3
"""Geometry fixtures from the reference's own host code (authoring container only).

For a set of cases built from the reference's *example* classes (examples/ldc_2d.py,
ldc_3d.py, poiseuille.py, poiseuille_3d.py) and its LBGeometryProcessor, records what
Subdomain.reset() + GeoEncoderConst produce: the encoded node map (uint32, ghosts
included), the un-encoded type map, the encoder's bit widths / type remap / parameter
table, _init_shape() sizes, and the initial host fields.  Output:
tests/golden/geometry_<case>.npz (+ geometry_cases.json describing each case so that the
tests can rebuild it with sailfish_amd).

    python tools/capture_geometry.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: F401
sys.path.insert(2, '/root/reference/examples')

import numpy as np

from sailfish.backend_dummy import DummyBackend
from sailfish.config import LBConfig
from sailfish.controller import LBGeometryProcessor
from sailfish.io import LBOutput
from sailfish import node_type as nt
from sailfish.subdomain_runner import SubdomainRunner

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


class _Logger(object):
    def debug(self, *a):
        pass
    info = warning = error = debug


def base_config(dim, **kw):
    c = LBConfig()
    c.init_iters = 0
    c.seed = 0
    c.precision = 'single'
    c.block_size = 64
    c.mem_alignment = 32
    c.node_addressing = 'direct'
    c.logger = _Logger()
    c.mode = 'batch'
    c.periodic_x = c.periodic_y = c.periodic_z = False
    c.use_link_tags = True
    c.time_dependence = False
    c.space_dependence = False
    c.access_pattern = 'AB'
    c.bulk_boundary_split = True
    c.output = ''
    c.max_iters = 10
    c.subdomains = 1
    c.conn_axis = 'x'
    c.grid = 'D2Q9' if dim == 2 else 'D3Q19'
    c.visc = 0.01
    c.model = 'bgk'
    c.incompressible = False
    c.minimize_roundoff = False
    c.relaxation_enabled = True
    c.propagation_enabled = True
    c.debug_dump_node_type_map = False
    c.debug_dump_dists = False
    c.check_invalid_results_gpu = False
    c.check_invalid_results_host = False
    c.output_format = 'npy'
    c.every = 100
    c.from_ = 0
    for k, v in kw.items():
        setattr(c, k, v)
    return c


CASES = {
    'ldc_2d': dict(module='ldc_2d', sim='LDCSim', dim=2, cfg=dict(lat_nx=40, lat_ny=24)),
    'ldc_3d': dict(module='ldc_3d', sim='LDCSim', dim=3, cfg=dict(lat_nx=20, lat_ny=14, lat_nz=12)),
    'ldc_3d_2sub_z': dict(module='ldc_3d', sim='LDCSim', dim=3, geo='EqualSubdomainsGeometry3D',
                          cfg=dict(lat_nx=20, lat_ny=14, lat_nz=12, subdomains=2, conn_axis='z')),
    'ldc_2d_3sub_x': dict(module='ldc_2d', sim='LDCSim', dim=2, geo='EqualSubdomainsGeometry2D',
                          cfg=dict(lat_nx=42, lat_ny=24, subdomains=3, conn_axis='x')),
    'poiseuille_force_fullbb': dict(module='poiseuille', sim='PoiseuilleSim', dim=2,
                                    cfg=dict(lat_nx=20, lat_ny=32, visc=0.1, horizontal=False, stationary=False,
                                             drive='force', wall='fullbb', force_implementation='guo')),
    'poiseuille_force_halfbb_h': dict(module='poiseuille', sim='PoiseuilleSim', dim=2,
                                      cfg=dict(lat_nx=32, lat_ny=20, visc=0.1, horizontal=True, stationary=True,
                                               drive='force', wall='halfbb', force_implementation='guo')),
    'poiseuille_pressure': dict(module='poiseuille', sim='PoiseuilleSim', dim=2,
                                cfg=dict(lat_nx=20, lat_ny=32, visc=0.1, horizontal=False, stationary=True,
                                         drive='pressure', wall='fullbb', force_implementation='guo')),
    'poiseuille_3d_force': dict(module='poiseuille_3d', sim='PoiseuilleSim', dim=3,
                                cfg=dict(lat_nx=18, lat_ny=18, lat_nz=10, visc=0.1, flow_direction='z',
                                         stationary=False, drive='force', wall='fullbb',
                                         force_implementation='guo')),
}


def run_case(name, case):
    import importlib
    mod = importlib.import_module(case['module'])
    importlib.reload(mod)
    sim_cls = getattr(mod, case['sim'])
    cfg = base_config(case['dim'], **case['cfg'])
    sim_cls.modify_config(cfg)
    from sailfish import geo as ref_geo
    geo_name = case.get('geo') or ('LBGeometry2D' if case['dim'] == 2 else 'LBGeometry3D')
    geo = getattr(ref_geo, geo_name)(cfg)
    specs = geo.subdomains()
    for s in specs:
        s.set_actual_size(1)
    specs = LBGeometryProcessor(specs, case['dim'], geo.gsize).transform(cfg)
    out = {}
    meta = {'n_subdomains': len(specs), 'locations': [list(s.location) for s in specs],
            'sizes': [list(s.size) for s in specs],
            'local_periodicity': [[bool(x) for x in s._periodicity] for s in specs],
            'face_conns': [sorted([int(f), int(i)] for f, i in s.connecting_subdomains()) for s in specs]}
    for spec in specs:
        sim = sim_cls(cfg)
        runner = SubdomainRunner(sim, spec, output=LBOutput(cfg, spec.id), backend=DummyBackend(), quit_event=None)
        runner._init_geometry()
        sim.init_fields(runner)
        runner._subdomain.init_fields(sim)
        sub = runner._subdomain
        ctx = {}
        sub.update_context(ctx)
        k = 's%d_' % spec.id
        out[k + 'encoded_map'] = np.array(sub._type_map_base, dtype=np.uint32)
        out[k + 'vis_map'] = np.array(sub._type_vis_map, dtype=np.uint8)
        out[k + 'physical_size'] = np.array(runner._physical_size)
        out[k + 'lat_size'] = np.array(runner._lat_size)
        out[k + 'node_params'] = np.array([float(x) for x in ctx['node_params']], dtype=np.float64)
        remap = ctx['type_id_remap']
        out[k + 'remap_keys'] = np.array(sorted(remap.keys()))
        out[k + 'remap_vals'] = np.array([remap[x] for x in sorted(remap.keys())])
        out[k + 'bits'] = np.array([ctx['nt_misc_shift'], ctx['nt_param_shift'], ctx['nt_scratch_shift']])
        out[k + 'rho'] = np.array(runner.field_base(sim.rho))
        for d, c in enumerate(sim.v):
            out[k + 'v%d' % d] = np.array(runner.field_base(c))
        out[k + 'num_fluid_nodes'] = np.array(sub.num_fluid_nodes)
    np.savez_compressed(os.path.join(OUT, 'geometry_%s.npz' % name), **out)
    return meta


def main():
    metas = {}
    for name, case in CASES.items():
        m = run_case(name, case)
        m.update({'module': case['module'], 'sim': case['sim'], 'dim': case['dim'], 'geo': case.get('geo'),
                  'cfg': case['cfg']})
        metas[name] = m
        print('captured', name, m['n_subdomains'], 'subdomain(s)')
    metas['_node_type_ids'] = dict((t.__name__, int(i)) for i, t in nt._NODE_TYPES.items())
    with open(os.path.join(OUT, 'geometry_cases.json'), 'w') as fh:
        json.dump(metas, fh, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
