"""One process per subdomain, started by the controller itself (reference sailfish/master.py:242-312
`_run_subprocesses`: one multiprocessing.Process per subdomain, round-robin over the GPUs of the machine,
master.py:106-117; controller.py:557-562 starts the machine master).

MI355X node: 8 GPUs on two CPU sockets, usually inside a container whose cgroup grants fewer CPUs than the box shows.
Every rank is a Python process that enqueues its GPU's steps (one C-ABI call per step once the step plans exist, see
stepqueue.py) and polls nothing, so one or two cores per rank are plenty -- but they should be cores of the NUMA node
its GPU hangs off, and the ranks must not all sit on the same few cores: `cpu_sets()` deals the allowed CPUs out per
GPU.  The ranks rendezvous through torch.distributed on 127.0.0.1 (RCCL when every rank has a GPU of its own, gloo with
host staging when ranks share one: functional runs on small boxes).
"""
import os
import socket
import sys


def free_port():
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(','):
        part = part.strip()
        if not part:
            continue
        if '-' in part:
            a, b = part.split('-')
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_numa_node(pci_bus_id, sysfs='/sys'):
    """NUMA node of the GPU with PCI address `pci_bus_id` ('0000:05:00.0'), or -1."""
    try:
        with open(os.path.join(sysfs, 'bus', 'pci', 'devices', pci_bus_id.lower(), 'numa_node')) as fh:
            return int(fh.read().strip())
    except (OSError, ValueError):
        return -1


def numa_cpus(node, sysfs='/sys'):
    try:
        with open(os.path.join(sysfs, 'devices', 'system', 'node', 'node%d' % node, 'cpulist')) as fh:
            return parse_cpulist(fh.read())
    except (OSError, ValueError):
        return []


def cpu_quota(cgroup='/sys/fs/cgroup/cpu.max'):
    """CPUs' worth of time the cgroup grants (None: unlimited / unknown)."""
    try:
        with open(cgroup) as fh:
            quota, period = fh.read().split()[:2]
        return None if quota == 'max' else max(1, int(int(quota) // int(period)))
    except (OSError, ValueError):
        return None


def cpu_sets(gpu_nodes, allowed, node_cpus, per_rank=None):
    """gpu_nodes[r]: NUMA node of rank r's GPU (-1 unknown); allowed: CPUs this process may run on; node_cpus(node) ->
    CPUs of that node.  Returns one CPU list per rank: ranks whose GPUs share a node share that node's allowed CPUs
    evenly (contiguous slices, `per_rank` CPUs at most), ranks of unknown nodes share whatever is left over.  No rank
    ends up with an empty set (falls back to all allowed CPUs)."""
    allowed = sorted(allowed)
    n = len(gpu_nodes)
    out = [None] * n
    by_node = {}
    for r, node in enumerate(gpu_nodes):
        by_node.setdefault(node, []).append(r)
    used = set()
    for node, ranks in sorted(by_node.items()):
        if node < 0:
            continue
        cpus = [c for c in node_cpus(node) if c in set(allowed)]
        if not cpus:
            by_node.setdefault(-1, []).extend(ranks)
            continue
        share = max(1, len(cpus) // len(ranks))
        if per_rank:
            share = min(share, per_rank)
        for i, r in enumerate(ranks):
            mine = cpus[i * share:(i + 1) * share] or cpus[-share:]
            out[r] = mine
            used.update(mine)
    rest = [c for c in allowed if c not in used] or allowed
    ranks = [r for r in by_node.get(-1, []) if out[r] is None]
    if ranks:
        share = max(1, len(rest) // len(ranks))
        if per_rank:
            share = min(share, per_rank)
        for i, r in enumerate(ranks):
            out[r] = rest[i * share:(i + 1) * share] or rest[-share:]
    return [o or allowed for o in out]


def plan_ranks(n_subdomains, gpus):
    """Rank r owns subdomain r on GPU gpus[r % len(gpus)] (reference master.py:106-117).  Returns (gpu per rank,
    process-group backend): 'nccl' (= RCCL) when every rank has a GPU of its own, 'gloo' when ranks share one."""
    per_rank = [int(gpus[r % len(gpus)]) for r in range(n_subdomains)]
    backend = 'nccl' if len(set(per_rank)) == len(per_rank) else 'gloo'
    return per_rank, backend


def picklable_config(cfg):
    """The parsed options as a plain dict a child process can take as its defaults."""
    import pickle
    out = {}
    for k, v in vars(cfg).items():
        if k == 'logger':
            continue
        try:
            pickle.dumps(v)
        except Exception:  # noqa: BLE001
            continue
        out[k] = v
    return out


def _pci_bus_ids(gpus):
    try:
        from sailfish_amd.backend_hip import HIPBackend
        return [HIPBackend.pci_bus_id(g) for g in gpus]
    except Exception:  # noqa: BLE001
        return [''] * len(gpus)


def rank_main(rank, world, port, gpu, dist_backend, cpus, lb_class, lb_geo, defaults, conn):
    """Body of one subdomain process: the controller's own torch.distributed branch with the environment a launcher
    would have set."""
    os.environ.update({'RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_RANK': str(gpu), 'SLF_FORCE_DEVICE': str(gpu),
                       'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'SLF_DIST_BACKEND': dist_backend,
                       'SLF_SPAWNED_RANK': '1'})
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    if dist_backend == 'gloo':
        # ranks that share a device: two hardware queues each, so that many processes do not oversubscribe the device's
        # queue slots (a counter hop of the peer transport: 13 us instead of 2.7 ms, profiles/r06/ipc_probe.txt)
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '2')
        os.environ.setdefault('SLF_HALO_PRIORITY', '0')     # a waiting kernel in a high-priority queue holds the other processes back
    if cpus and hasattr(os, 'sched_setaffinity'):
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    from sailfish_amd.controller import LBSimulationController
    ctrl = LBSimulationController(lb_class, lb_geo, default_config=defaults)
    res = ctrl.run(ignore_cmdline=True)
    if conn is not None:
        summary = None
        if getattr(ctrl, 'timing_infos', None) is not None and rank == 0:
            summary = {'timing': res[:3], 'mlups_total': getattr(ctrl, 'mlups_total', None),
                       'mlups_comp': getattr(ctrl, 'mlups_comp', None)}
        conn.send(summary)
        conn.close()
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


def run_processes(lb_class, lb_geo, cfg, n_subdomains, gpus, log=None):
    """Starts one process per subdomain and waits for them.  Returns rank 0's benchmark summary (or None)."""
    import multiprocessing as mp
    per_rank, backend = plan_ranks(n_subdomains, gpus)
    if backend == 'gloo' and log:
        log('several subdomains share a GPU: process group "gloo", halos staged through the host (functional, not fast)')
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(os.cpu_count() or 1))
    nodes = [gpu_numa_node(b) if b else -1 for b in _pci_bus_ids(per_rank)]
    quota = cpu_quota()
    share = max(1, quota // n_subdomains) if quota else None
    sets = cpu_sets(nodes, allowed, numa_cpus, per_rank=max(2, share) if share else None)
    if log:
        log('starting %d subdomain processes: gpus %s, numa nodes %s, cpu quota %s, %s cpus per rank'
            % (n_subdomains, per_rank, nodes, quota, [len(s) for s in sets]))
    ctx = mp.get_context('spawn')
    port = free_port()
    defaults = picklable_config(cfg)
    parent, child = ctx.Pipe(duplex=False)
    procs = []
    for r in range(n_subdomains):
        p = ctx.Process(target=rank_main, args=(r, n_subdomains, port, per_rank[r], backend, sets[r], lb_class, lb_geo,
                                                defaults, child if r == 0 else None), name='Subdomain/%d' % r)
        p.start()
        procs.append(p)
    child.close()
    summary, got = None, False
    try:
        # wait for rank 0's summary, but keep an eye on the others: a rank that died leaves its neighbours waiting in an
        # exchange for ever (the reference's master polls its subprocesses the same way, master.py:268-312)
        while True:
            if not got and parent.poll(0.2):
                try:
                    summary = parent.recv()
                except EOFError:
                    summary = None
                got = True
            failed = [p for p in procs if p.exitcode not in (None, 0)]
            if failed:
                for p in procs:
                    if p.exitcode is None:
                        p.terminate()
                break
            if all(p.exitcode is not None for p in procs):
                break
            if got:
                for p in procs:
                    p.join(0.2)
    finally:
        for p in procs:
            p.join(30)
            if p.exitcode is None:
                p.kill()
                p.join()
    bad = [(p.name, p.exitcode) for p in procs if p.exitcode != 0]
    if bad:
        raise RuntimeError('subdomain processes failed: %s' % bad)
    return summary


if __name__ == '__main__':
    print(cpu_sets([0, 0, 1, 1], range(16), lambda n: list(range(8 * n, 8 * n + 8))), file=sys.stderr)
