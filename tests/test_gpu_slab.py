"""The multi-slab (multi-GPU) driver of bench.py, exercised on ONE GPU: two SlabSim instances of a
2-rank ring live in this process and exchange their halo tensors through a loop-back exchanger
(device copies on the halo streams instead of RCCL send/recv).  Everything else -- face boxes,
pack / unpack kernels, boundary / bulk split, events, torch ExternalStream hand-over -- is the code
that runs under torch.distributed.  The merged result must equal the single-slab run of the whole
box bit for bit."""
import numpy as np
import pytest

from sailfish_amd import sym

pytestmark = pytest.mark.gpu


class Loopback(object):
    """Stands in for RingExchanger: collects the buffers of both ranks, then copies."""

    def __init__(self):
        self.bufs = {}

    def bind(self, rank):
        outer = self

        class _E(object):
            def exchange(self, send_up, send_down, recv_low, recv_high):
                outer.bufs[rank] = (send_up, send_down, recv_low, recv_high)
        return _E()


def step_pair(sims, lb):
    """One step of a 2-rank ring living in this process: rank r's send_up -> rank r+1's recv_low, send_down -> rank
    r-1's recv_high (2 ranks: the other one)."""
    import torch
    for s in sims:
        s.step_compute()
    for s in sims:
        s.step_exchange()
    for r, s in enumerate(sims):
        o = 1 - r
        s_up, s_down, _, _ = lb.bufs[r]
        _, _, r_low, r_high = lb.bufs[o]
        torch.cuda.synchronize()
        r_low.copy_(s_up)
        r_high.copy_(s_down)
    torch.cuda.synchronize()
    for s in sims:
        s.step_finish()


@pytest.mark.parametrize('pattern', ['AA', 'AB'])
@pytest.mark.parametrize('model,axis', [('bgk', 'z'), ('mrt', 'z'), ('bgk', 'y'), ('bgk', 'x'), ('mrt', 'x')])
def test_two_slabs_equal_one_box(pattern, model, axis):
    import torch
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.slab import AXES, SlabSim

    class Opt(object):
        pass
    n = (40, 12, 8)
    a = AXES[axis]
    lb = Loopback()
    # one backend object per rank, as in the real one-process-per-GPU run (the iteration counter of the
    # AA kernels is per backend)
    sims = [SlabSim(HIPBackend(Opt(), 0), sym.D3Q19, n, rank=r, world=2, model=model, access_pattern=pattern,
                    visc=0.02, exchanger=lb.bind(r), axis=axis) for r in range(2)]
    for s in sims:
        s.init_synthetic(seed=5)
    steps = 9
    for _ in range(steps):
        step_pair(sims, lb)
    np_axis = 3 - a                                   # arrays are [q, z, y, x]
    got = np.concatenate([s.real_view(s.get_dist()) for s in sims], axis=np_axis)

    whole = list(n)
    whole[a] *= 2
    one = SlabSim(HIPBackend(Opt(), 0), sym.D3Q19, tuple(whole), rank=0, world=1, model=model, access_pattern=pattern,
                  visc=0.02)
    # same initial state as the two slabs
    rho = np.concatenate([s.real_view(s.rho) for s in sims], axis=np_axis - 1)
    v = [np.concatenate([s.real_view(s.v[d]) for s in sims], axis=np_axis - 1) for d in range(3)]
    one.set_fields(rho, v)
    one.initial_conditions()
    for _ in range(steps):
        one.step()
    ref = one.real_view(one.get_dist())
    assert np.array_equal(got, ref)


@pytest.mark.parametrize('pattern', ['AA', 'AB'])
@pytest.mark.parametrize('axis', ['x', 'z'])
def test_single_slab_ring_of_one(pattern, axis):
    """world = 1 through the halo path (bench.py --force_distributed): the slab is its own ring neighbour, faces
    travel through the pack / exchange / unpack machinery instead of the in-sweep wrap.  Same populations."""
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.connector import RingExchanger
    from sailfish_amd.slab import SlabSim

    class Opt(object):
        pass
    n = (64, 10, 9)
    res = []
    for force in (True, False):
        s = SlabSim(HIPBackend(Opt(), 0), sym.D3Q19, n, rank=0, world=1, access_pattern=pattern, visc=0.02, axis=axis,
                    force_halo=force, exchanger=RingExchanger(0, 1) if force else None)
        s.init_synthetic(seed=3)
        for _ in range(8):
            s.step()
        s.sync()
        res.append(s.real_view(s.get_dist()))
    assert np.array_equal(res[0], res[1])


def _ring_of_one(axis, pattern, n, steps, env=None, model='bgk', restore_at=None):
    """Populations after `steps` steps of one slab that is its own ring neighbour (plain device copies: a transport a
    step plan can hold), under the environment switches `env`."""
    import os
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.connector import RingExchanger
    from sailfish_amd.slab import SlabSim

    class Opt(object):
        pass
    old = dict((k, os.environ.get(k)) for k in (env or {}))
    os.environ.update(env or {})
    try:
        s = SlabSim(HIPBackend(Opt(), 0), sym.D3Q19, n, rank=0, world=1, access_pattern=pattern, visc=0.02, axis=axis,
                    model=model, force_halo=True, exchanger=RingExchanger(0, 1))
        s.init_synthetic(seed=3)
        for i in range(steps):
            if restore_at is not None and i == restore_at:
                s.set_dist(s.get_dist())        # a state written from the host in the middle of the run
            s.step(save_macro=(i == steps - 1))
        s.sync()
        planned = sorted(s._plans)
        rho, v = s.fetch_fields()
        return s.real_view(s.get_dist()), s.real_view(rho).copy(), planned
    finally:
        for k, v_ in old.items():
            if v_ is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v_


@pytest.mark.parametrize('pattern', ['AA', 'AB'])
@pytest.mark.parametrize('axis', ['x', 'y', 'z'])
def test_step_plan_equals_direct_enqueue(pattern, axis):
    """The step replayed from a C-ABI plan (slf_plan_run), enqueued entry by entry from Python, and on one calc stream
    instead of two: the same populations and fields, bit for bit (four z-chunks along x)."""
    n = (64, 10, 40)
    ref, ref_rho, planned = _ring_of_one(axis, pattern, n, 9)
    assert planned == [(0, 0), (0, 1), (1, 0)]           # both parities, the last step with field output
    for env in ({'SLF_STEP_PLAN': '0'}, {'SLF_STEP_PLAN': '0', 'SLF_CALC_STREAMS': '1'}, {'SLF_CALC_STREAMS': '1'},
                {'SLF_XFACE_CHUNKS': '2'}, {'SLF_XFACE_CHUNKS': '1'}, {'SLF_XFACE_STREAMS': '2'}, {'SLF_XFACE_BATCHES': 'two'},
                {'SLF_XFACE_CHUNKS': '8'}):
        got, got_rho, planned = _ring_of_one(axis, pattern, n, 9, env)
        assert (planned == []) == (env.get('SLF_STEP_PLAN') == '0')
        assert np.array_equal(got, ref), env
        assert np.array_equal(got_rho, ref_rho), env


@pytest.mark.parametrize('restore_at', [3, 4])
def test_x_slab_state_written_at_odd_iteration(restore_at):
    """a state written from the host (checkpoint restore, set_dist) at an ODD in-place iteration: the next
    step pulls, and the edge lanes of the fluid-only row kernel do not look into the ghost columns -- the receive
    buffers are primed from them (XFaceHalo.prime_pull)."""
    n = (64, 10, 16)
    ref, _, _ = _ring_of_one('x', 'AA', n, 8)
    got, _, _ = _ring_of_one('x', 'AA', n, 8, restore_at=restore_at)
    assert np.array_equal(got, ref)


def test_step_plan_c_abi_basics():
    """slf_plan_*: entries are performed in order by one call; kernels that take the iteration get it from
    slf_plan_run(); misuse is refused when the entry is added, not when the plan runs."""
    import ctypes
    from sailfish_amd.backend_hip import HIPBackend, HIPEvent, HIPKernel
    from sailfish_amd.box import BoxSim, make_box_desc

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    desc = make_box_desc(sym.D3Q19, (32, 8, 6), access_pattern='AA', visc=0.02, periodic_fused=[1, 1, 1])
    ref = BoxSim(b, desc, periodic=(True, True, True))
    sim = BoxSim(b, desc, periodic=(True, True, True))
    rng = np.random.RandomState(0)
    rho = 1.0 + 1e-3 * rng.rand(6, 8, 32)
    v = [0.01 * rng.rand(6, 8, 32) for _ in range(3)]
    for s in (ref, sim):
        s.set_fields(rho, v)
        s.initial_conditions()
    for _ in range(4):
        ref.step()
    plan = b.make_plan()
    ev = HIPEvent(b)
    other = b.make_stream()
    plan.launch(sim.k_sweep[0][0], None, sim.stream)          # the AA kernel: parity from the iteration given to run()
    plan.record(ev, sim.stream)
    plan.wait(other, ev)
    plan.memset(sim.gpu_rho, 0, 16, other)
    assert len(plan) == 4
    for it in range(4):
        plan.run(it)
    sim.iteration = 4
    sim.sync()
    other.synchronize()
    assert np.array_equal(sim.real_view(sim.get_dist()), ref.real_view(ref.get_dist()))
    unbound = HIPKernel(b._lib, sim.module, 'CollideAndPropagate')
    with pytest.raises(b.FatalError):
        plan.launch(unbound, None, sim.stream)
    assert b._lib.slf_plan_run(None, 0) != 0
    assert b._lib.slf_plan_add_memset(plan.handle, ctypes.c_void_p(sim.gpu_rho), 0, 16, None) != 0      # a stream is needed
    assert len(plan) == 4
