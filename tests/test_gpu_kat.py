"""The reference's propagation known-answer tests (see tests/test_kat_oracle.py) on the GPU, through
the full stack exactly as the reference runs them: LBSimulationController with --debug_dump_dists
--every=1, expected slots read back from the dumped <output>_dists.<subdomain>.<iteration>.npz files."""
import numpy as np
import pytest

from sailfish_amd import io
from tests import _kat

pytestmark = pytest.mark.gpu

RUNS = [(i, r) for i, r in _kat.all_runs() if _kat.supported(r)]


@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'ghostpbc'])
@pytest.mark.parametrize('rid,run', RUNS, ids=[i for i, _ in RUNS])
def test_propagation_kat_gpu(rid, run, fused, tmp_path):
    from sailfish_amd.controller import LBSimulationController
    sim_cls, geo_cls, cfg, grid = _kat.make_classes(run)
    out = str(tmp_path / 'kat')
    cfg.update(hip_fused_periodic=fused, output=out, every=1, debug_dump_dists=True, quiet=True,
               perf_stats_every=0, check_invalid_results_host=False)
    # relaxation_enabled is an internal option (set by modify_config in the reference tests)
    relax = cfg.pop('relaxation_enabled')
    sim_cls.modify_config = classmethod(lambda cls, config: setattr(config, 'relaxation_enabled', relax))
    LBSimulationController(sim_cls, geo_cls, default_config=cfg).run(ignore_cmdline=True)
    digits = io.filename_iter_digits(run['max_iters'])
    cache = {}

    def get(sid, it):
        if (sid, it) not in cache:
            cache[(sid, it)] = np.load(io.dists_filename(out, digits, sid, it))['arr_0']
        return cache[(sid, it)]
    _kat.check(run, grid, get)
