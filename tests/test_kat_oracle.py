"""The reference's exact propagation known-answer tests (regtest/subdomains/2d_propagation.py:121-705,
3d_propagation.py:150-747: tagged values written into single slots, streamed 1-5 steps without
relaxation across block / subdomain faces, edges, corners, global and mixed periodic boundaries, AB and
AA access patterns) replayed on the CPU: oracle kernels + the product's geometry and halo routing."""
import numpy as np
import pytest

from tests import _kat
from tests._oracle_group import OracleSubdomain
from tests import _host

RUNS = [(i, r) for i, r in _kat.all_runs() if _kat.supported(r)]


@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'ghostpbc'])
@pytest.mark.parametrize('rid,run', RUNS, ids=[i for i, _ in RUNS])
def test_propagation_kat(rid, run, fused):
    sim_cls, geo_cls, cfg, grid = _kat.make_classes(run)
    cfg['hip_fused_periodic'] = fused
    cfg_, specs, runners = _host.build_runners(sim_cls, run['dim'], geo_cls, cfg)
    subs = [OracleSubdomain(r) for r in runners]
    for s in subs:
        for d in s.dist:
            s.raw(d)[:] = 0.0
            for inp in run['inputs']:
                if inp['subdomain'] == s.runner._spec.id:
                    idx = (grid.vec_idx(inp['vec']),) + ((0,) if run['dim'] == 2 else ()) + tuple(inp['pos'])
                    d[idx] = inp['value']
    dumps = {}
    for it in range(1, run['max_iters'] + 1):
        sends = [s.compute() for s in subs]
        for s in subs:
            s.finish(dict((nid, sends[nid][s.runner._spec.id]) for nid in s.links))
        for s in subs:
            cur = s.dist[0] if s.aa else s.dist[s.iteration & 1]
            dumps[(s.runner._spec.id, it)] = cur[:, 0].copy() if run['dim'] == 2 else cur.copy()
    _kat.check(run, grid, lambda sid, it: dumps[(sid, it)])
