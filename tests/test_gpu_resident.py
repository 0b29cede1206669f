"""Several time steps inside one launch for launch-bound 2-D subdomains (library kernel CollideAndPropagateResident,
csrc/slf_resident.hip; SubdomainRunner._fast_forward_resident): the arrays after N steps are, slot for slot -- ghost
layer, never-written slots and both copies of the two-copy pattern included -- what one launch per step leaves behind
(`bit-identical to stepping`).  The reference launches CollideAndPropagate once per step
(subdomain_runner.py:960-974); its AB == AA check (tests/gpu/access_pattern.sh) is repeated here across the two paths."""
import numpy as np
import pytest

from tests import _open_sims
from tests.test_gpu_runner import run_gpu

pytestmark = pytest.mark.gpu

CASES = {
    # walls (full-way bounce-back) + the regularized-velocity lid: BASELINE config 1's geometry
    'cavity': ('ldc_2d', 'LDCSim', dict(lat_nx=48, lat_ny=40, visc=0.02)),
    'cavity_mrt': ('ldc_2d', 'LDCSim', dict(lat_nx=45, lat_ny=52, visc=0.03, model='mrt')),
    'cavity_f64': ('ldc_2d', 'LDCSim', dict(lat_nx=40, lat_ny=33, visc=0.02, precision='double')),
    # x wrapped inside the sweep (the window wraps too), body force, walls along y
    'channel': ('poiseuille', 'PoiseuilleSim', dict(lat_nx=36, lat_ny=30, visc=0.05, horizontal=True, drive='force', wall='fullbb')),
    # full-slip walls (boundary-condition nodes of the window's list), x wrapped, body force
    'slip_channel': (_open_sims.SlipChannelSim, None, dict(lat_nx=36, lat_ny=30, visc=0.05, periodic_x=True,
                                                           force_implementation='guo')),
    # the configuration itself
    'cavity_256': ('ldc_2d', 'LDCSim', dict(lat_nx=256, lat_ny=256, visc=0.0254)),
}


def _state(r):
    copies = [0, 1] if r.config.access_pattern == 'AB' else [0]
    return [r._debug_get_dist(copy=c) for c in copies] + [np.array(r._sim.rho), np.array(r._sim.v[0]), np.array(r._sim.v[1])]


@pytest.mark.parametrize('case', sorted(CASES))
@pytest.mark.parametrize('pattern', ['AA', 'AB'])
@pytest.mark.parametrize('steps,every', [(75, 37), (301, 301)])
def test_resident_steps_equal_plain_stepping(case, pattern, steps, every, tmp_path, monkeypatch):
    monkeypatch.setenv('SLF_RESIDENT_FORCE', '1')       # wherever the kernel applies, not only where it pays (double precision)
    module, sim, cfg = CASES[case]
    if case == 'cavity_256' and every == 37:
        pytest.skip('one long stretch is enough at this size')
    res = {}
    for resident in (True, False):
        extra = dict(hip_resident=resident, hip_graphs=resident, every=every)
        ctrl = run_gpu(module, sim, 2, dict(cfg, access_pattern=pattern), steps, extra=extra)
        r = ctrl.runners[0]
        assert r._sim.iteration == steps
        res[resident] = _state(r)
        if resident:
            assert isinstance(r._resident, dict) and r._resident['graphs'], 'the resident path was not taken'
            per_launch = r._resident['steps']
            assert per_launch == {'AA': 8, 'AB': 7}[pattern] and r._resident['halo'] == 8
            if every == 301:      # 300 host-free steps: graphs of 16 and of 2 launches
                assert sorted(set(k[0] for k in r._resident['graphs'])) == [2, 16]
        else:
            assert not r._resident
    for a, b in zip(res[True], res[False]):
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize('pattern,taken', [('AB', True), ('AA', False)])
def test_resident_steps_with_a_do_nothing_outlet(pattern, taken, monkeypatch):
    """An open channel with NTDoNothing on its last column.  Two-copy pattern: those are fluid nodes that read slots nobody
    ever writes -- the window is a cache of the raw slots, so the resident steps leave what stepping leaves.  In place the
    nodes store into memory from their node code (the ghost column behind them): the library refuses the kernel and the
    runner steps."""
    monkeypatch.setenv('SLF_RESIDENT_FORCE', '1')
    res = {}
    for resident in (True, False):
        ctrl = run_gpu(_open_sims.OpenChannelSim, None, 2, dict(lat_nx=48, lat_ny=36, visc=0.05, access_pattern=pattern), 75,
                       extra=dict(hip_resident=resident, hip_graphs=resident, every=75))
        r = ctrl.runners[0]
        res[resident] = _state(r)
        assert bool(r._resident) == (resident and taken)
    for a, b in zip(res[True], res[False]):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize('cfg,taken', [(dict(lat_nx=256, lat_ny=256), True), (dict(lat_nx=40, lat_ny=33), True),
                                       (dict(lat_nx=40, lat_ny=33, precision='double'), False), (dict(lat_nx=512, lat_ny=512), False)])
def test_resident_path_is_taken_where_it_pays(cfg, taken):
    """The runner's own choice (no SLF_RESIDENT_FORCE): single precision and at most 90 000 nodes -- a 512^2 cavity runs 23.8
    GMLUPS with the resident kernel and 33.6 launch by launch, double precision 7.2 against 11.0 at 256^2
    (profiles/r05/ldc2d_resident_variants.txt)."""
    ctrl = run_gpu('ldc_2d', 'LDCSim', 2, dict(cfg, visc=0.0254, access_pattern='AA'), 40, extra=dict(every=40))
    r = ctrl.runners[0]
    assert bool(r._resident) == taken and r._sim.iteration == 40


def test_resident_kernel_is_refused_where_it_does_not_apply():
    """Half-way bounce-back nodes write memory from their node code: the library refuses the kernel, the runner keeps
    stepping (and the results are what they were)."""
    cfg = dict(lat_nx=36, lat_ny=30, visc=0.05, horizontal=True, drive='force', wall='halfbb', access_pattern='AB')
    res = {}
    for resident in (True, False):
        ctrl = run_gpu('poiseuille', 'PoiseuilleSim', 2, cfg, 60, extra=dict(hip_resident=resident, every=60))
        r = ctrl.runners[0]
        assert not r._resident
        res[resident] = _state(r)
    for a, b in zip(res[True], res[False]):
        assert np.array_equal(a, b, equal_nan=True)


def test_resident_kernel_argument_checks():
    from sailfish_amd import sym
    from sailfish_amd.backend_hip import HIPBackend, HIPFatalError
    from sailfish_amd.box import BoxSim, make_box_desc

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    desc = make_box_desc(sym.D2Q9, (40, 30), precision='single', access_pattern='AA', visc=0.02, periodic_fused=[1, 1, 0])
    sim = BoxSim(b, desc, periodic=(True, True, False))
    scratch = b.alloc_buf(size=9 * sim.stride * 4, align_offset=b.dist_align_offset(4))
    d = sim.gpu_dist[0]

    def get(args, fmt='PPPPPiiiii', it=True):
        return b.get_kernel(sim.module, 'CollideAndPropagateResident', (64,), args, fmt, needs_iteration=it)
    get([0, d, 0, scratch, 0, 0, 8, 4, 4, 8])
    for bad, what in (([0, d, 0, d, 0, 0, 8, 4, 4, 8], 'different buffers'), ([0, d, 0, scratch, 0, 0, 8, 4, 4, 7], 'halo too small'),
                      ([0, d, 0, scratch, 0, 0, 8, 40, 40, 8], 'larger than 2048'), ([0, d, 0, scratch, 0, 0, 0, 4, 4, 8], 'positive')):
        with pytest.raises(HIPFatalError, match=what):
            get(bad)
    with pytest.raises(HIPFatalError, match='iteration'):
        get([0, d, 0, scratch, 0, 0, 8, 4, 4, 8], it=False)
    desc3 = make_box_desc(sym.D3Q19, (20, 12, 10), precision='single', access_pattern='AA', visc=0.02, periodic_fused=[1, 1, 1])
    sim3 = BoxSim(b, desc3, periodic=(True, True, True))
    with pytest.raises(HIPFatalError, match='2-D'):
        b.get_kernel(sim3.module, 'CollideAndPropagateResident', (64,), [0, sim3.gpu_dist[0], 0, scratch, 0, 0, 8, 4, 4, 8],
                     'PPPPPiiiii', needs_iteration=True)


@pytest.mark.parametrize('pattern', ['AA', 'AB'])
def test_resident_box_equals_the_oracle(pattern):
    """The kernel by itself on a periodic fluid-only box (no node map: the non-GENERAL instantiation, both axes wrapped)
    against the CPU oracle: populations bit for bit after 3 launches."""
    from sailfish_amd import sym
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.box import BoxSim, make_box_desc
    from tests._oracle_box import OracleBox, synthetic_fields

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    size = (50, 34)
    desc = make_box_desc(sym.D2Q9, size, precision='single', access_pattern=pattern, visc=0.03, periodic_fused=[1, 1, 0])
    rho, v = synthetic_fields(size, 2)
    g, o = BoxSim(b, desc, periodic=(True, True, False)), OracleBox(desc, periodic=(True, True, False))
    for s in (g, o):
        s.set_fields(rho, v)
        s.initial_conditions()
    steps = 8 if pattern == 'AA' else 7
    isz = 4
    nbytes = 9 * g.stride * isz
    scratch = [b.alloc_buf(size=nbytes, align_offset=b.dist_align_offset(isz)) for _ in g.gpu_dist]
    src = list(g.gpu_dist) + [0] * (2 - len(g.gpu_dist))
    dst = scratch + [0] * (2 - len(scratch))
    for a, c in zip(g.gpu_dist, scratch):
        b.copy_buf_async(c, a, nbytes, g.stream)
    ints = [0, steps, 7, 5, 8]
    fwd = b.get_kernel(g.module, 'CollideAndPropagateResident', (64,), [0] + src + dst + ints, 'PPPPPiiiii', needs_iteration=True)
    bwd = b.get_kernel(g.module, 'CollideAndPropagateResident', (64,), [0] + dst + src + ints, 'PPPPPiiiii', needs_iteration=True)
    it = 0
    for k in (fwd, bwd, fwd):
        b.set_iteration(it)
        b.run_kernel(k, None, g.stream)
        it += steps
    for a, c in zip(g.gpu_dist, scratch):       # the third launch left the state in the scratch copies
        b.copy_buf_async(a, c, nbytes, g.stream)
    g.stream.synchronize()
    g.iteration = it
    b.set_iteration(it)
    o.run(it, save_last=False)
    assert np.array_equal(g.real_view(g.get_dist()), o.real_view(o.current_dist()))
    # ... and stepping on from there works
    g.run(3, save_last=True)
    o.run(3, save_last=True)
    assert np.array_equal(g.real_view(g.get_dist()), o.real_view(o.current_dist()))
