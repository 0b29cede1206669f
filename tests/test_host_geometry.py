"""Host geometry layer vs fixtures produced by the reference's own host code
(tests/golden/geometry_*.npz, tools/capture_geometry.py): node-type ids, subdomain
connectivity, padded sizes (reference tests/subdomain_runner.py:54-77), unused /
propagation-only detection and orientation / link tags (tests/subdomain.py:98-576),
the GeoEncoderConst bit packing, and the initial host fields -- all bit-exact.

Each case is built twice: from this repo's examples/ (written against the same API), and --
where /root/reference is available -- from the reference's *unchanged* example file imported
against the `sailfish` alias package.
"""
import json
import os

import numpy as np
import pytest

from sailfish_amd import node_type as nt
from tests import _host

with open(os.path.join(os.path.dirname(__file__), 'golden', 'geometry_cases.json')) as fh:
    CASES = json.load(fh)
NAMES = sorted(k for k in CASES if not k.startswith('_'))
HAVE_REF = os.path.isdir('/root/reference/examples')


def test_node_type_ids_match_reference():
    ref = CASES['_node_type_ids']
    mine = dict((t.__name__, i) for i, t in nt._NODE_TYPES.items())
    assert mine == ref


@pytest.mark.parametrize('source', ['repo_examples', 'reference_files'])
@pytest.mark.parametrize('name', NAMES)
def test_geometry_matches_reference(name, source, golden_dir):
    if source == 'reference_files' and not HAVE_REF:
        pytest.skip('/root/reference not available on this machine')
    case = CASES[name]
    G = np.load(os.path.join(golden_dir, 'geometry_%s.npz' % name))
    sim_cls = _host.load_sim_class(case['module'], case['sim'], use_reference_file=(source == 'reference_files'))
    cfg, specs, runners = _host.build_runners(sim_cls, case['dim'], case['geo'], case['cfg'])
    assert len(specs) == case['n_subdomains']
    for spec, runner in zip(specs, runners):
        i = spec.id
        assert list(spec.location) == case['locations'][i] and list(spec.size) == case['sizes'][i]
        assert [bool(x) for x in spec._periodicity] == case['local_periodicity'][i]
        assert sorted([int(f), int(n)] for f, n in spec.connecting_subdomains()) == case['face_conns'][i]
        runner._init_geometry()
        runner._sim.init_fields(runner)
        runner._subdomain.init_fields(runner._sim)
        k = 's%d_' % i
        assert runner._physical_size == list(G[k + 'physical_size'])
        assert runner._lat_size == list(G[k + 'lat_size'])
        sub = runner._subdomain
        assert np.array_equal(sub._type_vis_map, G[k + 'vis_map'])
        ctx = {}
        sub.update_context(ctx)
        assert [ctx['nt_misc_shift'], ctx['nt_param_shift'], ctx['nt_scratch_shift']] == list(G[k + 'bits'])
        remap = ctx['type_id_remap']
        assert sorted(remap.keys()) == list(G[k + 'remap_keys'])
        assert [remap[x] for x in sorted(remap.keys())] == list(G[k + 'remap_vals'])
        assert np.allclose(np.array(ctx['node_params'], dtype=np.float64), G[k + 'node_params'], rtol=0, atol=0)
        enc = np.array(sub._type_map_base, dtype=np.uint32)
        assert np.array_equal(enc, G[k + 'encoded_map']), \
            '%d differing node codes' % np.count_nonzero(enc != G[k + 'encoded_map'])
        assert sub.num_fluid_nodes == int(G[k + 'num_fluid_nodes'])
        assert np.array_equal(runner.field_base(runner._sim.rho), G[k + 'rho'])
        for d, c in enumerate(runner._sim.v):
            assert np.array_equal(runner.field_base(c), G[k + 'v%d' % d])


def test_padded_sizes_known_answers():
    """reference tests/subdomain_runner.py:54-77: (10,3) -> [3,16] and (3,5,7) -> [7,5,8] at alignment 8."""
    from sailfish_amd.lb_single import LBFluidSim
    from sailfish_amd.subdomain import SubdomainSpec2D, SubdomainSpec3D
    from sailfish_amd.subdomain_runner import SubdomainRunner
    cfg = _host.make_config(2, mem_alignment=8, lat_nx=64, lat_ny=64)
    spec = SubdomainSpec2D((0, 0), (10, 3))
    spec.set_actual_size(0)
    r = SubdomainRunner(LBFluidSim(cfg), spec, None, _host.HostOnlyBackend())
    r._init_shape()
    assert r._physical_size == [3, 16] and r.num_phys_nodes == 48
    cfg = _host.make_config(3, mem_alignment=8, lat_nx=64, lat_ny=64, lat_nz=64)
    spec = SubdomainSpec3D((0, 0, 0), (3, 5, 7))
    spec.set_actual_size(0)
    r = SubdomainRunner(LBFluidSim(cfg), spec, None, _host.HostOnlyBackend())
    r._init_shape()
    assert r._physical_size == [7, 5, 8] and r.num_phys_nodes == 280


def test_weighted_subdomains_balance_active_nodes(tmp_path):
    """WeightedSubdomainsGeometry3D (reference geo.py:137-176): slabs with about the same number of active
    nodes; falls back to equal slabs without a decomposition geometry."""
    from sailfish_amd.geo import EqualSubdomainsGeometry3D, WeightedSubdomainsGeometry3D
    from tests._host import make_config
    nx, ny, nz = 40, 12, 10
    inactive = np.ones((nz, ny, nx), dtype=bool)
    # a cone-like channel: wide at low x, narrow at high x
    for x in range(nx):
        w = max(1, int(5 * (1.0 - x / float(nx)) ** 2) + 1)
        inactive[nz // 2 - w:nz // 2 + w, ny // 2 - w:ny // 2 + w, x] = False
    fn = str(tmp_path / 'geo.npy')
    np.save(fn, inactive)
    cfg = make_config(3, lat_nx=nx, lat_ny=ny, lat_nz=nz, subdomains=4, conn_axis='x', geometry_for_decomposition=fn)
    specs = WeightedSubdomainsGeometry3D(cfg).subdomains()
    assert len(specs) == 4 and specs[0].location[0] == 0 and specs[-1].end_location[0] == nx
    for a, b in zip(specs[:-1], specs[1:]):
        assert a.end_location[0] == b.location[0]            # contiguous, no overlap
    counts = [int((~inactive[:, :, s.location[0]:s.end_location[0]]).sum()) for s in specs]
    total = int((~inactive).sum())
    assert sum(counts) == total
    assert max(counts) - min(counts) <= 0.35 * total / 4       # balanced to within a layer of the wide end
    equal = [s.size[0] for s in EqualSubdomainsGeometry3D(cfg).subdomains()]
    assert [s.size[0] for s in specs] != equal and specs[0].size[0] < specs[-1].size[0]
    cfg.geometry_for_decomposition = ''
    assert [s.size[0] for s in WeightedSubdomainsGeometry3D(cfg).subdomains()] == equal


REF_EXAMPLES = [
    ('sc_phase_separation', 'SCSim', 2, {}),
    ('sc_drop', 'SCSim', 2, {}),
    ('binary_fluid/sc_separation_2d', 'SeparationSCSim', 2, {}),
    ('binary_fluid/sc_separation_3d', 'SeparationSCSim', 3, {'lat_nx': 24, 'lat_ny': 20, 'lat_nz': 16}),
    ('binary_fluid/sc_capillary', None, 2, {}),                 # body forces on both lattices
    ('binary_fluid/sc_capillary_wave_2d', None, 2, {}),         # force_implementation = edm
    ('binary_fluid/sc_drop_2d', None, 2, {}),
    ('binary_fluid/sc_laplace_2d', None, 2, {}),
    ('binary_fluid/sc_poiseuille_2d', None, 2, {}),
    ('binary_fluid/sc_rayleigh_taylor_2d', None, 2, {}),
    ('cylinder', None, 2, {}),
    ('duct_flow', None, 3, {'lat_nx': 32, 'lat_ny': 32, 'lat_nz': 32}),
    ('sphere_3d', None, 3, {'lat_nx': 64, 'lat_ny': 32, 'lat_nz': 32}),
    ('taylor_green_2d', None, 2, {}),
    ('external_geometry', 'ExternalSimulation', 3, {}),
    ('external_geometry', 'ExternalSimulation', 3, {'node_addressing': 'indirect'}),
]


@pytest.mark.skipif(not HAVE_REF, reason='/root/reference not available on this machine')
@pytest.mark.parametrize('module,sim,dim,extra', REF_EXAMPLES)
def test_more_reference_examples_load_unchanged(module, sim, dim, extra):
    """The reference's own example files (Shan-Chen models, external geometry incl. indirect addressing),
    imported unchanged against the `sailfish` alias package: options parse, the subdomain geometry and the
    initial fields build, and the module descriptor for the HIP backend comes out (no GPU involved)."""
    import importlib.util
    import sailfish  # noqa: F401
    from sailfish_amd import hipabi
    path = os.path.join('/root/reference/examples', module + '.py')
    spec = importlib.util.spec_from_file_location('refexample_' + module.replace('/', '_'), path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from sailfish_amd.lb_base import LBSim
    if sim is None:      # the simulation class the example defines
        import inspect
        sim_cls = [c for c in vars(mod).values()
                   if inspect.isclass(c) and issubclass(c, LBSim) and c.__module__ == mod.__name__][-1]
    else:
        sim_cls = getattr(mod, sim)
    assert issubclass(sim_cls, LBSim)
    defaults = {}
    sim_cls.update_defaults(defaults)
    cfg = dict(defaults)
    cfg.update(extra)
    if module == 'external_geometry':
        cfg['geometry'] = 'pipe.npy'
    geo = 'LBGeometry%dD' % dim
    # every option at its declared default (the example's own add_options included), as the command line
    # parser of the controller would deliver them
    from sailfish_amd import geo as geo_mod
    from sailfish_amd.controller import LBSimulationController
    ctrl = LBSimulationController(sim_cls, getattr(geo_mod, geo), default_config=cfg)
    parsed = ctrl._config_parser.parse([])
    full = dict((k, v) for k, v in vars(parsed).items() if not k.startswith('_'))
    full.update(cfg)
    cfg_, specs, runners = _host.build_runners(sim_cls, dim, geo, full)
    r = runners[0]
    r._init_geometry()
    r._sim.init_fields(r)
    r._subdomain.init_fields(r._sim)
    desc = r._module_desc()
    assert desc.lat_nx == cfg_.lat_nx + 2
    assert np.isfinite(r._sim.rho).all() and r._subdomain.num_fluid_nodes > 0
    if extra.get('node_addressing') == 'indirect':
        assert desc.node_addressing == hipabi.SLF_ADDR_INDIRECT
        assert r._subdomain.active_nodes < 0.8 * np.prod(r._subdomain.full_lat_shape)
    if 'sc_' in module and 'binary_fluid' not in module or sim in ('SeparationSCSim',):
        assert desc.simtype in (hipabi.SLF_SIM_SHAN_CHEN_BINARY, hipabi.SLF_SIM_SHAN_CHEN_SINGLE)


def test_x_slabs_are_cut_on_line_boundaries_where_the_balance_allows(monkeypatch):
    """geo._split_rows: cuts along x at multiples of 32 nodes (a 128-byte line of single-precision values) when every
    slab stays within 15 % of the equal share; the reference's equal pieces otherwise, always for --slab_align=0, and along
    y / z (reference geo.py:113-135)."""
    from sailfish_amd import geo
    assert geo._split_rows(512, 3, 32) == [(0, 160), (160, 192), (352, 160)]
    assert geo._split_rows(512, 3, 0) == geo._split(512, 3) == [(0, 170), (170, 170), (340, 172)]
    assert geo._split_rows(1024, 8, 32) == geo._split(1024, 8)                 # already on lines
    assert geo._split_rows(512, 5, 32) == geo._split(512, 5)                   # 96 / 128 would be 25 % off the share
    assert geo._split_rows(100, 3, 32) == geo._split(100, 3)                   # slabs narrower than two lines
    assert geo._split_rows(1000, 3, 32) == [(0, 320), (320, 352), (672, 328)]  # the last slab takes the odd end
    for total, parts in ((512, 3), (1000, 3), (2048, 3), (777, 2), (4096, 7)):
        pieces = geo._split_rows(total, parts, 32)
        assert pieces[0][0] == 0 and all(a + n == b for (a, n), (b, _) in zip(pieces, pieces[1:]))
        assert sum(n for _, n in pieces) == total and all(abs(n - total / parts) <= 0.15 * total / parts for _, n in pieces)

    class Cfg(object):
        lat_nx, lat_ny, lat_nz, subdomains = 512, 64, 48, 3
    for axis, align, want in (('x', 32, [160, 192, 160]), ('x', 0, [170, 170, 172]), ('y', 32, [21, 21, 22]), ('z', 32, [16, 16, 16])):
        Cfg.conn_axis, Cfg.slab_align = axis, align
        specs = geo.EqualSubdomainsGeometry3D(Cfg).subdomains()
        assert [s.size['xyz'.index(axis)] for s in specs] == want
    # the defaults: slabs of one process on ONE device run one after the other -- whole waves matter, balance does not;
    # a device per slab (or a rank of several): lines, within 15 % of the equal share
    Cfg.conn_axis, Cfg.slab_align = 'x', None
    for gpus, world, want in ((0, None, [192, 128, 192]), ([0], None, [192, 128, 192]), ([0, 1, 2], None, [160, 192, 160]),
                              ([0, 0, 0], None, [160, 192, 160]), ([0], '3', [160, 192, 160]), (None, None, [160, 192, 160])):
        Cfg.gpus = gpus
        if world is None:
            monkeypatch.delenv('WORLD_SIZE', raising=False)
        else:
            monkeypatch.setenv('WORLD_SIZE', world)
        assert [s.size[0] for s in geo.EqualSubdomainsGeometry3D(Cfg).subdomains()] == want
    assert geo._split_rows(512, 3, 64, 0.3) == [(0, 192), (192, 128), (320, 192)]
