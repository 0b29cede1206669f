"""Worker of the world_size-2 gloo test: one subdomain per process, halo buffers exchanged with the
product's TorchDistConnector (CPU tensors over gloo); arithmetic by the oracle."""
import os
import sys

import numpy as np


def worker(rank, world, port, case, steps, outdir):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from sailfish_amd.connector import TorchDistConnector, init_distributed
    from tests import _host
    from tests._oracle_group import OracleSubdomain
    r, w = init_distributed('gloo')
    assert (r, w) == (rank, world)
    module, sim, dim, geo, cfg = case
    sim_cls = _host.load_sim_class(module, sim)
    cfg_, specs, runners = _host.build_runners(sim_cls, dim, geo, cfg)
    assert len(specs) == world
    sub = OracleSubdomain(runners[rank])
    conn = TorchDistConnector(dict((s.id, s.id) for s in specs), device=torch.device('cpu'))
    for i in range(steps):
        sends = sub.compute(save=(i == steps - 1))
        counts = sub.recv_counts()
        nids = sorted(sub.links)
        tdt = torch.float32 if sub.o.dtype == np.float32 else torch.float64
        s_list = [(torch.from_numpy(sends[n]), conn.id_to_rank[n]) for n in nids if len(sends[n])]
        r_bufs = dict((n, torch.empty(counts[n], dtype=tdt)) for n in nids)
        r_list = [(r_bufs[n], conn.id_to_rank[n]) for n in nids if counts[n]]
        conn.exchange_tensors(s_list, r_list)
        sub.finish(dict((n, r_bufs[n].numpy()) for n in nids))
    cur = sub.dist[0] if sub.aa else sub.dist[sub.iteration & 1]
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), dist=np.ascontiguousarray(sub.real(cur)),
             rho=np.ascontiguousarray(sub.real(sub.rho)), location=np.array(specs[rank].location),
             size=np.array(specs[rank].size))
    dist.barrier()
    dist.destroy_process_group()


def ring_worker(rank, world, port, outdir):
    """RingExchanger (the exchange of bench.py --gpus N) on CPU tensors over gloo."""
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from sailfish_amd.connector import RingExchanger, init_distributed
    r, w = init_distributed('gloo')
    ex = RingExchanger(r, w)
    n = 1000
    got = []
    for step in range(3):
        send_up = torch.full((n,), 100.0 * r + 10.0 * step + 1.0)       # -> rank + 1, its recv_low
        send_down = torch.full((n,), 100.0 * r + 10.0 * step + 2.0)     # -> rank - 1, its recv_high
        recv_low, recv_high = torch.zeros(n), torch.zeros(n)
        ex.exchange(send_up, send_down, recv_low, recv_high)
        got.append((float(recv_low[0]), float(recv_low[-1]), float(recv_high[0]), float(recv_high[-1])))
    np.save(os.path.join(outdir, 'ring%d.npy' % rank), np.array(got))
    dist.barrier()
    dist.destroy_process_group()


def nn_worker(rank, world, port, dim, size, nsub_axis, single, steps, outdir):
    """Shan-Chen models, one subdomain per process: two exchanges per step (macroscopic fields, then the
    populations of every lattice) through TorchDistConnector.exchange_tensors over gloo."""
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from sailfish_amd.connector import TorchDistConnector, init_distributed
    from tests import _host, _sc
    from tests._oracle_group import OracleSCSingle, OracleSCSubdomain
    init_distributed('gloo')
    sim_cls, _ = (_sc.make_single_sim if single else _sc.make_sim)(dim)
    cfg = (_sc.single_config if single else _sc.config)(dim, size, pattern='AA')
    if single:
        cfg.update(G=-1.2, sc_potential='linear')
    cfg.update(subdomains=world, conn_axis=nsub_axis)
    cfg_, specs, runners = _host.build_runners(sim_cls, dim, 'EqualSubdomainsGeometry%dD' % dim, cfg)
    sub = (OracleSCSingle if single else OracleSCSubdomain)(runners[rank])
    conn = TorchDistConnector(dict((s.id, s.id) for s in specs), device=torch.device('cpu'))
    tdt = torch.float32 if sub.o.dtype == np.float32 else torch.float64

    def swap(sends, counts):
        nids = sorted(sends)
        s_list = [(torch.from_numpy(np.ascontiguousarray(sends[n])), conn.id_to_rank[n]) for n in nids if len(sends[n])]
        bufs = dict((n, torch.empty(counts[n], dtype=tdt)) for n in nids)
        r_list = [(bufs[n], conn.id_to_rank[n]) for n in nids if counts[n]]
        conn.exchange_tensors(s_list, r_list)
        return dict((n, bufs[n].numpy()) for n in nids)

    nf, nl = len(sub.nn_fields()), (1 if single else 2)
    for i in range(steps):
        sends = sub.macro_send()
        sub.macro_recv(swap(sends, dict((n, len(l.recv) * nf) for n, l in sub.macro_links.items())))
        sends = sub.dist_send(i == steps - 1)
        sub.dist_recv(swap(sends, dict((n, len(getattr(l, sub._mode + '_recv')) * nl) for n, l in sub.links.items())))
    cur = sub.current() if single else sub.current()[0]
    np.savez(os.path.join(outdir, 'nn%d.npz' % rank), dist=np.ascontiguousarray(sub.real(cur)),
             rho=np.ascontiguousarray(sub.real(sub.rho)), location=np.array(specs[rank].location),
             size=np.array(specs[rank].size))
    dist.barrier()
    dist.destroy_process_group()


def controller_worker(rank, world, port, case, steps, outdir):
    """The PRODUCT's process-per-subdomain path on the CPU: LBSimulationController.run() under WORLD_SIZE > 1 ->
    init_distributed -> SubdomainRunner.run() -> step() -> halo_messages() -> TorchDistConnector.exchange(runner)
    (reference subdomain_runner.py:1028-1139), with tests/_oracle_backend.OracleBackend standing in for the GPU
    (kernels executed by the oracle on host memory, halo tensors on the CPU, gloo instead of RCCL)."""
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from sailfish_amd import geo as geo_mod, util
    from sailfish_amd.controller import LBSimulationController
    from tests import _host
    from tests._oracle_backend import OracleBackend
    on_gpu = os.environ.get('SLF_TEST_CONTROLLER_ON_GPU') == '1'     # test_gpu_two_ranks.py: the real HIP backend
    if not on_gpu:
        util.get_backends = lambda backends=('hip',): iter([OracleBackend])
    module, sim, dim, geo, cfg = case
    sim_cls = _host.load_sim_class(module, sim)
    cfg = dict(cfg, max_iters=steps, quiet=True, perf_stats_every=0, backends='hip' if on_gpu else 'oracle_test')
    ctrl = LBSimulationController(sim_cls, getattr(geo_mod, geo), default_config=cfg)
    ctrl.run(ignore_cmdline=True)
    assert dist.is_initialized() and dist.get_world_size() == world and dist.get_backend() == 'gloo'
    assert len(ctrl.runners) == 1
    r = ctrl.runners[0]
    assert r._spec.id == rank and r._links, 'the runner of this rank must have halo links'
    # on the GPU the ranks share the device and can map each other's memory: the peer transport (unless the test asks for
    # host staging with SLF_HALO_TRANSPORT=torch); the CPU test backend has nothing to map
    want = 'PeerConnector' if (on_gpu and os.environ.get('SLF_HALO_TRANSPORT', 'auto') in ('auto', 'peer')) else 'TorchDistConnector'
    assert type(r._connector).__name__ == want and r._sim.iteration == steps, type(r._connector).__name__
    f = r._debug_get_dist()
    sl = (slice(None),) + tuple(r._spec._nonghost_slice)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), dist=np.ascontiguousarray(f[sl]),
             rho=np.ascontiguousarray(r._sim.rho), location=np.array(r._spec.location), size=np.array(r._spec.size))
    dist.barrier()
    dist.destroy_process_group()
