"""SubdomainRunner._steps_without_host: how many of the coming steps may be replayed as HIP graphs because
the host has nothing to do in them (no output, field transfer, checkpoint, statistics line, user hook, end
of run).  Pure host logic: checked against the step-by-step predicates of the main loop."""
import pytest

from tests import _host


def _runner(**cfg):
    base = dict(lat_nx=16, lat_ny=12, visc=0.05, max_iters=100, every=10, from_=0, output='', checkpoint_every=0,
                checkpoint_file='', perf_stats_every=0, mode='batch')
    base.update(cfg)
    sim_cls = _host.load_sim_class('ldc_2d', 'LDCSim')
    cfg_, specs, runners = _host.build_runners(sim_cls, 2, 'LBGeometry2D', base)
    return runners[0]


def _brute_force(r):
    """Steps from the current iteration until one that needs the host, using the main loop's own predicates."""
    sim, cfg = r._sim, r.config
    it0 = sim.iteration
    n = 0
    try:
        while True:
            it = it0 + n
            sim.iteration = it
            last = cfg.max_iters > 0 and it + 1 >= cfg.max_iters
            if last or sim.need_output():
                return n
            sim.iteration = it + 1                       # state after the step
            if sim.need_checkpoint() and cfg.checkpoint_file:
                return n
            if cfg.perf_stats_every > 0 and (it + 1) % cfg.perf_stats_every == 0:
                return n
            n += 1
            if n > 10000:
                return n
    finally:
        sim.iteration = it0


@pytest.mark.parametrize('cfg', [
    dict(),
    dict(output='/tmp/x', every=10),
    dict(output='/tmp/x', every=7, from_=23),
    dict(checkpoint_every=13, checkpoint_file='/tmp/c'),
    dict(perf_stats_every=25),
    dict(output='/tmp/x', every=10, checkpoint_every=15, checkpoint_file='/tmp/c', perf_stats_every=9),
    dict(max_iters=0, output='/tmp/x', every=50),
])
def test_matches_the_main_loop_predicates(cfg):
    r = _runner(**cfg)
    for it in range(0, 99):
        r._sim.iteration = it
        expect = _brute_force(r)
        got = r._steps_without_host()
        assert got == min(expect, got) and (got == expect or expect > 10000), (cfg, it, got, expect)


def test_user_hooks_and_requests_disable_it():
    r = _runner()
    assert r._steps_without_host() == 99
    r._sim.need_sync_flag = True
    assert r._steps_without_host() == 0
    r._sim.need_sync_flag = False
    r._checkpoint_req = True
    assert r._steps_without_host() == 0
    r._checkpoint_req = False

    class Hooked(type(r._sim)):
        def after_step(self, runner):
            pass
    r._sim.__class__ = Hooked
    assert r._steps_without_host() == 0


def test_benchmark_sampling_starts_exactly_at_sample_from():
    r = _runner(mode='benchmark', max_iters=3000, benchmark_sample_from=1000, benchmark_minibatch=50)
    from sailfish_amd.profile import TimeProfile
    r._profile = TimeProfile(r)
    r._sim.iteration = 990
    assert r._steps_without_host() == 10
    r._sim.iteration = 1000
    assert r._steps_without_host() == 1999
