"""Option handling of the host layer (reference config.py:33-91, controller.py:441-455): options are
collected from the simulation class, its subdomain, the geometry class and the backend; defaults come from
update_defaults(), rc files, default_config and the command line, in that order of precedence."""
import os

from sailfish_amd import geo as geo_mod
from sailfish_amd.controller import LBSimulationController
from tests import _host


def _ctrl(defaults=None):
    sim_cls = _host.load_sim_class('poiseuille', 'PoiseuilleSim')
    return LBSimulationController(sim_cls, geo_mod.EqualSubdomainsGeometry2D, default_config=defaults)


def test_options_from_every_layer_and_precedence(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    cfg = _ctrl()._config_parser.parse([])
    # model layer, example, geometry, backend, controller
    assert cfg.visc == 0.1 and cfg.model == 'bgk' and cfg.grid == 'D2Q9'      # 0.1: the example's update_defaults
    assert cfg.drive == 'force' and cfg.subdomains == 1 and cfg.conn_axis == 'x'
    assert cfg.hip_fused_periodic is True and cfg.hip_graphs is True
    assert cfg.access_pattern == 'AB' and cfg.node_addressing == 'direct' and cfg.precision == 'single'
    assert cfg.check_invalid_results_gpu is True and cfg.force_implementation == 'guo'
    # default_config beats update_defaults, the command line beats both
    cfg = _ctrl({'visc': 0.05, 'lat_nx': 40})._config_parser.parse(['--visc=0.02', '--access_pattern=AA'])
    assert cfg.visc == 0.02 and cfg.lat_nx == 40 and cfg.access_pattern == 'AA'
    assert cfg.needs_iteration_num and not cfg.output_required
    cfg = _ctrl()._config_parser.parse(['--output=/tmp/x', '--nohip_graphs', '--nocheck_invalid_results_gpu'])
    assert cfg.output_required and cfg.hip_graphs is False and cfg.check_invalid_results_gpu is False


def test_rc_file(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    with open(os.path.join(str(tmp_path), '.sailfishrc'), 'w') as f:
        f.write('[main]\nprecision = double\n')
    cfg = _ctrl()._config_parser.parse([])
    assert cfg.precision == 'double'
    cfg = _ctrl()._config_parser.parse(['--precision=single'])
    assert cfg.precision == 'single'


def test_minimize_roundoff_is_a_density_model_and_refused_where_it_does_not_apply():
    """--minimize_roundoff (reference lb_base.py:72-76, sym.py:573-661): the third value of the module's density
    model; BGK with fluid and bounce-back nodes -- everything else is refused, never silently run as the standard
    formulation."""
    import pytest
    from sailfish_amd import hipabi
    from sailfish_amd.controller import LBSimulationController
    from sailfish_amd.geo import LBGeometry2D
    from sailfish_amd.lb_single import LBFluidSim
    from tests import _host
    cfg = _host.make_config(2, minimize_roundoff=True)
    assert LBFluidSim.density_model(cfg) == hipabi.SLF_DENSITY_ROUNDOFF
    assert LBFluidSim.density_model(_host.make_config(2)) == hipabi.SLF_DENSITY_COMPRESSIBLE
    assert LBFluidSim.density_model(_host.make_config(2, incompressible=True)) == hipabi.SLF_DENSITY_INCOMPRESSIBLE
    with pytest.raises(ValueError):
        LBFluidSim.density_model(_host.make_config(2, minimize_roundoff=True, model='mrt'))
    ok = dict(incompressible=hipabi.SLF_DENSITY_ROUNDOFF, type_kind=[hipabi.SLF_NK_FLUID, hipabi.SLF_NK_GHOST, hipabi.SLF_NK_FULL_BB])
    LBFluidSim.check_module_desc(ok)
    with pytest.raises(NotImplementedError):       # the lid of the cavity examples
        LBFluidSim.check_module_desc(dict(ok, type_kind=ok['type_kind'] + [hipabi.SLF_NK_REGULARIZED_VELOCITY]))
    # the geometry of examples/ldc_2d.py is refused on the host, before anything touches the device
    sim_cls = _host.load_sim_class('ldc_2d', 'LDCSim')
    _, _, runners = _host.build_runners(sim_cls, 2, LBGeometry2D, dict(minimize_roundoff=True, lat_nx=32, lat_ny=32))
    r = runners[0]
    r._init_geometry()
    r._sim.init_fields(r)
    with pytest.raises(NotImplementedError):
        r._module_desc()
    # binary models: refused by the controller
    from tests import _sc
    sc_cls, geo = _sc.make_sim(2)
    ctrl = LBSimulationController(sc_cls, geo, default_config=dict(_sc.config(2, (16, 16)), minimize_roundoff=True,
                                                                  max_iters=1, quiet=True))
    with pytest.raises(NotImplementedError):
        ctrl.run(ignore_cmdline=True)


def test_regularized_and_subgrid_options_reach_the_module_descriptor():
    """--regularized / --subgrid=les-smagorinsky / --smagorinsky_const (reference lb_single.py:27-42): same names and
    defaults, carried into slf_module_desc; MRT is refused on the host (the reference's MRT relaxation never calls the
    preamble that implements them and would ignore the flags)."""
    import pytest
    from sailfish_amd import hipabi
    from sailfish_amd.geo import LBGeometry2D
    from tests import _host
    sim_cls = _host.load_sim_class('ldc_2d', 'LDCSim')
    for opts, want in ((dict(), (0, hipabi.SLF_SUBGRID_NONE)),
                       (dict(regularized=True), (1, hipabi.SLF_SUBGRID_NONE)),
                       (dict(subgrid='les-smagorinsky', smagorinsky_const=0.17), (0, hipabi.SLF_SUBGRID_LES_SMAGORINSKY))):
        _, _, runners = _host.build_runners(sim_cls, 2, LBGeometry2D, dict(lat_nx=32, lat_ny=32, **opts))
        r = runners[0]
        r._init_geometry()
        r._sim.init_fields(r)
        d = r._module_desc()
        assert (int(d.regularized), int(d.subgrid)) == want
        if opts.get('subgrid'):
            assert d.smagorinsky_const == 0.17
    import argparse
    from sailfish_amd.lb_single import LBFluidSim
    ap = argparse.ArgumentParser()
    LBFluidSim.add_options(ap, 2)
    cfg = ap.parse_args([])
    assert cfg.regularized is False and cfg.subgrid == 'none' and cfg.smagorinsky_const == 0.1       # the reference's defaults
    _, _, runners = _host.build_runners(sim_cls, 2, LBGeometry2D, dict(lat_nx=32, lat_ny=32, regularized=True, model='mrt'))
    r = runners[0]
    r._init_geometry()
    r._sim.init_fields(r)
    with pytest.raises(ValueError, match='BGK'):
        r._module_desc()
