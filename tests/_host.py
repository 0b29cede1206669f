"""Host-side test doubles (the reference tests the host layer the same way: tests/common.py uses a
no-op DummyBackend + DummyLogger).  Nothing here computes LBM results."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class DummyLogger(object):
    def debug(self, *a):
        pass
    info = warning = error = debug


class HostOnlyBackend(object):
    """Enough of the backend interface to build geometry and host fields without a GPU."""
    name = 'hostonly'
    gpu_id = 0

    def alloc_async_host_buf(self, shape, dtype):
        return np.zeros(shape, dtype=dtype)


def make_config(dim, **kw):
    from sailfish_amd.config import LBConfig
    c = LBConfig()
    c.precision = 'single'
    c.block_size = 64
    c.mem_alignment = 32
    c.node_addressing = 'direct'
    c.logger = DummyLogger()
    c.mode = 'batch'
    c.periodic_x = c.periodic_y = c.periodic_z = False
    c.use_link_tags = True
    c.access_pattern = 'AB'
    c.bulk_boundary_split = True
    c.output = ''
    c.max_iters = 10
    c.subdomains = 1
    c.conn_axis = 'x'
    c.gpus = 0                  # the controller's default: one device (geo._row_split_rule cuts x-slabs accordingly)
    c.grid = 'D2Q9' if dim == 2 else 'D3Q19'
    c.visc = 0.01
    c.model = 'bgk'
    c.incompressible = False
    c.relaxation_enabled = True
    c.hip_fused_periodic = True
    c.every = 100
    c.from_ = 0
    c.checkpoint_every = 0
    c.checkpoint_from = 0
    c.checkpoint_file = ''
    c.perf_stats_every = 0
    for k, v in kw.items():
        setattr(c, k, v)
    return c


# reference example (module, sim class) -> this repo's example written against the same API
EXAMPLE_MAP = {
    ('ldc_2d', 'LDCSim'): ('examples.ldc_2d', 'CavitySim'),
    ('ldc_3d', 'LDCSim'): ('examples.ldc_3d', 'CavitySim'),
    ('poiseuille', 'PoiseuilleSim'): ('examples.poiseuille', 'ChannelSim'),
    ('poiseuille_3d', 'PoiseuilleSim'): ('examples.poiseuille_3d', 'PipeSim'),
    ('external_geometry', 'ExternalSimulation'): ('examples.external_geometry', 'GeometrySim'),
    ('cylinder', 'CylinderSimulation'): ('examples.cylinder', 'CylinderSim'),
    ('sphere_3d', 'SphereSimulation'): ('examples.sphere_3d', 'SphereSim'),
    ('womersley', 'WomersleySim'): ('examples.womersley', 'WomersleySim'),
    ('poiseuille_pulsatile', 'PulsatileSim'): ('examples.poiseuille_pulsatile', 'PulsatileSim'),
}


def load_sim_class(module, sim, use_reference_file=False):
    """use_reference_file: import the *reference's own example file* (unchanged) against the
    `sailfish` alias package -- only possible where /root/reference exists."""
    import sailfish  # noqa: F401  (installs the sailfish.* aliases)
    if use_reference_file:
        path = os.path.join('/root/reference/examples', module + '.py')
        spec = importlib.util.spec_from_file_location('refexample_' + module, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return getattr(mod, sim)
    mname, cname = EXAMPLE_MAP[(module, sim)]
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    mod = importlib.import_module(mname)
    importlib.reload(mod)
    return getattr(mod, cname)


def build_runners(sim_cls, dim, geo_name, cfg_kw, backend_factory=None):
    """Config -> subdomain specs -> one prepared-geometry runner per subdomain."""
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBGeometryProcessor
    from sailfish_amd.io import LBOutput
    cfg = make_config(dim, **cfg_kw)
    sim_cls.modify_config(cfg)
    if isinstance(geo_name, type):
        geo = geo_name(cfg)
    else:
        geo = getattr(geo_mod, geo_name or ('LBGeometry2D' if dim == 2 else 'LBGeometry3D'))(cfg)
    specs = geo.subdomains()
    for s in specs:
        s.set_actual_size(1)
    specs = LBGeometryProcessor(specs, dim, geo.gsize).transform(cfg)
    periodic = [cfg.periodic_x, cfg.periodic_y] + ([cfg.periodic_z] if dim == 3 else [])
    runners = []
    for spec in specs:
        sim = sim_cls(cfg)
        backend = backend_factory() if backend_factory else HostOnlyBackend()
        r = sim.subdomain_runner(sim, spec, LBOutput(cfg, spec.id), backend, None)
        r.set_topology(specs, geo.gsize, periodic)
        runners.append(r)
    return cfg, specs, runners
