#!/usr/bin/env python3
"""Headline benchmark: MLUPS of the fused collide-and-stream sweep, D3Q19 BGK, single precision, periodic box
(BASELINE.json metric; SURVEY.md §8(d) "M0" / "C4"), through backend_hip -> libsailfish_hip.so.

    python bench.py --gpus N --steps K --warmup W
        weak scaling (default): 512^3 per GPU, the global box is 512 x 512 x (512 N) cut into N z-slabs
    python bench.py --gpus N --scaling strong --domain 1024x512x512 --axis {x,z}
        BASELINE config 4: a fixed box cut into N slabs along x (the reference's default axis, geo.py:100-135) or z

N > 1: launched by torch.distributed.run, one rank per GPU; face layers are exchanged device-to-device over RCCL
every step on a halo stream that overlaps the interior sweep (sailfish_amd/slab.py).  --force_distributed takes the
same path with a single rank (the slab is its own ring neighbour; RCCL send / recv to self).

Prints ONE JSON line on rank 0 (field list: DESIGN.md §6).  `value` is the MEDIAN of the timed blocks of exactly K
steps (each bracketed by barrier + synchronize, max over ranks; blocks are repeated until --min_seconds are timed, for
--repeats fresh placements); `best_value` is the fastest block (what rounds 3-4 reported as `value`: not directly
comparable), every block is in `config`.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_UPDATE = {4: 152, 8: 304}   # D3Q19: 2 * Q * sizeof(real), SURVEY.md §8(d)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--size', type=int, default=512, help='weak scaling: box edge per GPU (512 = the headline config)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--domain', default='1024x512x512', help='strong scaling: the global box NXxNYxNZ')
    ap.add_argument('--axis', default='z', choices=['x', 'y', 'z'], help='axis the box is cut along')
    ap.add_argument('--force_distributed', action='store_true',
                    help='one rank, but through the halo path (pack, RCCL send / recv to self, unpack)')
    ap.add_argument('--access_pattern', default='auto', choices=['auto', 'AA', 'AB'],
                    help='AA = in-place single copy, AB = two copies (the reference default); auto times both '
                         'with the same K steps and reports the faster one')
    ap.add_argument('--model', default='bgk', choices=['bgk', 'mrt'])
    ap.add_argument('--precision', default='single', choices=['single', 'double'])
    ap.add_argument('--no_fused_periodic', action='store_true',
                    help='use ghost-layer PBC kernels (reference scheme) instead of in-sweep wrap')
    ap.add_argument('--visc', type=float, default=1.0 / 6.0)
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--cpu_seconds', type=float, default=10.0)
    ap.add_argument('--repeats', type=int, default=2,
                    help='each candidate access pattern is timed this many times (K steps each)')
    ap.add_argument('--no_placement_tune', action='store_true',
                    help='place the distribution arrays once instead of by measurement (sailfish_amd/placement.py choose())')
    ap.add_argument('--prewarm_steps', type=int, default=300,
                    help='untimed steps before the W warm-up steps (same count on every rank): about 1 s at 512^3')
    ap.add_argument('--no_gpu_state', action='store_true', help='do not sample amd-smi before / after')
    ap.add_argument('--no_runner_path', action='store_true',
                    help='skip the leg through LBSimulationController / SubdomainRunner.step() (N = 1 only)')
    ap.add_argument('--min_seconds', type=float, default=0.5,
                    help='the block of exactly K timed steps is repeated until at least this much time has been timed '
                         '(every block bracketed by barrier + synchronize; median block = value, the best one is reported too)')
    ap.add_argument('--halo_timing_steps', type=int, default=20,
                    help='N > 1: steps of the separate leg that brackets the halo stream with timing events')
    ap.add_argument('--no_validate', action='store_true',
                    help='skip the check of the final state against the CPU oracle after the timed region')
    return ap.parse_args()


def gpu_state():
    """Clocks, power and temperatures from `amd-smi metric` (evidence that a run was not throttled)."""
    try:
        out = subprocess.run(['amd-smi', 'metric', '-g', '0'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             timeout=20).stdout.decode(errors='replace')
    except Exception:  # noqa: BLE001
        return None
    st = {}
    gfx = [int(x) for x in re.findall(r'GFX_\d+:\s*\n\s*CLK:\s*(\d+) MHz', out)]
    if gfx:
        st['gfx_clk_mhz_min'], st['gfx_clk_mhz_max'] = min(gfx), max(gfx)
    for key, pat in (('mem_clk_mhz', r'MEM_0:\s*\n\s*CLK:\s*(\d+) MHz'), ('socket_power_w', r'SOCKET_POWER:\s*(\d+) W'),
                     ('temp_hotspot_c', r'HOTSPOT:\s*(\d+)'), ('temp_mem_c', r'\n\s*MEM:\s*(\d+)\s*.C'),
                     ('throttle', r'THROTTLE_STATUS:\s*(\S+)')):
        m = re.search(pat, out)
        if m:
            st[key] = int(m.group(1)) if m.group(1).isdigit() else m.group(1)
    return st or None


def cpu_baseline(args):
    """The CPU restatement of the sweep timed on this box's host cores: the blocked OpenMP twin
    (oracle/lbm_fast.c, bit-identical to the table-driven oracle: tests/test_cpu_twin.py) at 256^3 D3Q19 and on
    BASELINE config 1 (256^2 D2Q9), best of 3, all cores and one thread.  The reference has no CPU compute path
    (SURVEY.md F1), hence kind = "port"."""
    from oracle import cpu_twin
    return cpu_twin.baseline(args.model, args.precision, args.visc, budget_s=args.cpu_seconds)


def load_traffic(workload_key, path=None):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC runs (profiles/traffic.json), if one exists for this
    exact workload AND for these kernels: the file carries the sha256 of sailfish_amd/csrc/* it was measured with
    (sailfish_amd/build.py source_hash(); tools/traffic_update.py writes both) -- after any change of the kernel sources
    the line says `traffic: null` until the PMC passes have been repeated."""
    from sailfish_amd import build as slf_build
    p = path or os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(p) as fh:
            data = json.load(fh)
    except Exception:  # noqa: BLE001
        return None
    if data.get('_csrc_sha256') != slf_build.source_hash():
        return None
    return data.get(workload_key)


def median_block(blocks):
    """Index of the median timed block (the lower of the two middle ones for an even count: always a block that ran)."""
    order = np.argsort(np.asarray(blocks, dtype=np.float64), kind='stable')
    return int(order[(len(order) - 1) // 2])


def rccl_ranks_of(exchanger, dist_backend, world, peer=None):
    """How many ranks RCCL itself reports for the communicator the halos travel over (C ABI slf_comm_count ->
    ncclCommCount); 0 when they do not travel over RCCL (peer transport, gloo / host staging, plain copies).  With
    SLF_HALO_TRANSPORT=torch on an RCCL process group the communicator is torch's: its world size."""
    if peer is not None:
        return 0
    if exchanger is not None and getattr(exchanger, 'direct', False):
        return int(exchanger.rccl.count()[0])
    if dist_backend == 'nccl':
        return int(world)
    return 0


def halo_transport_of(sim):
    """How the halo of `sim` travels, in words (the line's `halo_transport`)."""
    from sailfish_amd import peer as peer_mod
    if getattr(sim, 'peer', None) is not None:
        return ('peer: the sweep / the pack kernels store into the neighbours\' receive buffers (HIP IPC mappings), ordered by '
                'progress counters (slf_peer_signal / slf_peer_wait inside the step plan); no copies, no RCCL kernels')
    ex = getattr(sim, 'exchanger', None)
    why = ' (peer transport unavailable: %s)' % peer_mod.unavailable_reason if peer_mod.unavailable_reason else ''
    if ex is not None and getattr(ex, 'direct', False):
        return 'RCCL through the C ABI (slf_comm_exchange inside the step plan)' + why
    if ex is not None and ex.plain_copy():
        return 'device copies (a ring of one without a process group)' + why
    return 'torch.distributed' + why


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one per GPU, the way the reference's
    controller spawns one process per subdomain (master.py:242-312) -- here through torch.distributed.run, the same
    command line the driver uses; rank 0 prints the JSON line.  Returns the exit status."""
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    if env.get('SLF_FORCE_DEVICE') is not None:
        # the ranks share ONE device: with the runtime's default of four hardware queues per process eight processes
        # oversubscribe the device's queue slots, and a counter hop of the peer transport costs 2.7 ms instead of 13 us
        # (profiles/r06/ipc_probe.txt)
        env.setdefault('GPU_MAX_HW_QUEUES', '2')
        # ... and no stream priorities: a high-priority queue with a waiting kernel in it holds the other processes'
        # sweeps back (eight z-slab processes: 29.5 GMLUPS with, 35.4 without, profiles/r06/peer_eight_processes.txt)
        env.setdefault('SLF_HALO_PRIORITY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def runner_path(args, gpu_id):
    """The same 512^3 periodic BGK box through the product's host stack -- LBSimulationController -> SubdomainRunner.step()
    with --mode=benchmark accounting, which is where the reference measures its MLUPS (subdomain_runner.py:1693-1703,
    controller.py:740-765) and what `examples run unchanged` exercises.  bench.py's `value` times the slab driver
    (sailfish_amd/slab.py); the two must agree."""
    import io as _io
    from contextlib import redirect_stdout
    from sailfish.controller import LBSimulationController
    from sailfish.geo import LBGeometry3D
    from sailfish.lb_single import LBFluidSim
    from sailfish.subdomain import Subdomain3D

    class PeriodicBox(Subdomain3D):
        def boundary_conditions(self, hx, hy, hz):
            pass

        def initial_conditions(self, sim, hx, hy, hz):
            sim.rho[:] = 1.0
            sim.vx[:] = 0.05 * np.sin(2 * np.pi * hy / self.gy)
            sim.vy[:] = 0.05 * np.sin(2 * np.pi * hz / self.gz)
            sim.vz[:] = 0.05 * np.sin(2 * np.pi * hx / self.gx)

    class BoxSim(LBFluidSim):
        subdomain = PeriodicBox

    n = args.size
    steps = max(200, args.steps + 100)
    cfg = dict(mode='benchmark', quiet=True, perf_stats_every=0, lat_nx=n, lat_ny=n, lat_nz=n, periodic_x=True,
               periodic_y=True, periodic_z=True, visc=args.visc, access_pattern=args.access_pattern, grid='D3Q19',
               model=args.model, precision=args.precision, max_iters=steps, benchmark_sample_from=100, gpus=[gpu_id])
    ctrl = LBSimulationController(BoxSim, LBGeometry3D, default_config=cfg)
    with redirect_stdout(_io.StringIO()):
        ctrl.run(ignore_cmdline=True)
    r = ctrl.runners[0]
    out = {'mlups': round(ctrl.mlups_total, 1), 'mlups_sweep_only': round(ctrl.mlups_comp, 1),
           'kernel_ms': round(r.num_fluid_nodes / ctrl.mlups_comp * 1e-3, 4), 'steps': steps,
           'access_pattern': args.access_pattern, 'placement_tuning': getattr(r, 'placement_tuning', None),
           'through': 'LBSimulationController -> SubdomainRunner.step(), --mode=benchmark'}
    r.release()
    return out


def ring_swapper(rank, world):
    """swap(low, high) -> (from_down, from_up) for oracle.window.SeamCheck: the edge layers of the ring neighbours, moved
    with the same point-to-point pattern as the halo itself (connector.RingExchanger: torch.distributed isend / irecv,
    RCCL for device tensors; a plain copy for a ring of one without a process group)."""
    import torch
    from sailfish_amd.connector import RingExchanger

    def flatten(tree):
        if isinstance(tree, (list, tuple)):
            return [a for t in tree for a in flatten(t)]
        return [tree]

    def rebuild(tree, flat):
        if isinstance(tree, (list, tuple)):
            return [rebuild(t, flat) for t in tree]
        return flat.pop(0)

    def swap(low, high):
        ex = RingExchanger(rank, world)
        fl, fh = flatten(low), flatten(high)
        dev = torch.device('cuda', torch.cuda.current_device())
        t_low = torch.from_numpy(np.concatenate([a.ravel() for a in fl])).to(dev)
        t_high = torch.from_numpy(np.concatenate([a.ravel() for a in fh])).to(dev)
        r_down, r_up = torch.empty_like(t_high), torch.empty_like(t_low)
        ex.exchange(t_high, t_low, r_down, r_up)      # my high layers travel up and arrive as the neighbour's "from down"
        torch.cuda.synchronize()
        out = []
        for flat_t, like in ((r_down, fh), (r_up, fl)):
            host, parts, o = flat_t.cpu().numpy(), [], 0
            for a in like:
                parts.append(host[o:o + a.size].reshape(a.shape).copy())
                o += a.size
            out.append(parts)
        return rebuild(high, out[0]), rebuild(low, out[1])
    return swap


def validate(sim, backend, mass0, distributed, axis, rank=0, world=1):
    """AFTER the timed region, on the state the timed steps left behind: (a) sampled z-planes of the next two steps
    against the CPU oracle (7-plane windows seeded from the device, oracle/window.py: bit-identical populations
    expected) -- for a slab with ring neighbours the windows reach three layers into the NEIGHBOURS' slabs
    (window.SeamCheck: their edge layers are fetched over the same point-to-point pattern as the halo), so the seam
    layers -- planes z = 1, z = nz of a z-slab, columns x = 1, x = nx of an x-slab -- are compared bit for bit;
    (b) total mass and momentum against the initial state (periodic BGK box without forcing: conserved up to f32
    round-off).  The oracle is the checker here, never part of what is timed."""
    from oracle import window
    from sailfish_amd.slab import AXES
    import torch
    out = {}
    nz = sim.desc.lat_nz - 2
    fields = [sim.gpu_rho] + list(sim.gpu_v)
    sim.sync()
    if not sim.halo:
        zs = [1, nz // 3, nz]
        zs += [z for z in window.chunk_boundary_planes(sim.placed, sim.desc, sim.stride, limit=2) if z not in zs]
        chk = window.PlaneCheck(backend, sim.desc, None, zs, sim.gpu_dist, sim.stride, fields)
    else:
        sim.materialise_faces()         # x-face buffers: what crossed the faces goes into the arrays
        zs = sorted(set([1, nz // 2, nz] if axis == 'z' else [1, nz // 3, nz]))
        if axis == 'z' and world >= 4:
            zs = [1, nz]          # the seam planes; a z-slab's windows span the whole x-y extent of the box (the mid plane
            #                       is covered by the undivided-box windows below)
        chk = window.SeamCheck(backend, sim.desc, zs, sim.gpu_dist, sim.stride, fields, AXES[axis], ring_swapper(rank, world))
        out['seam_layers_checked'] = ({'z': 'planes z = 1 and z = nz', 'y': 'rows y = 1 and y = ny',
                                       'x': 'columns x = 1 and x = nx'}[axis] + ' of every sampled plane (windows reach 3 layers '
                                      'into the ring neighbours)')
    glob = None
    if sim.halo and world > 1:
        # whole planes of the UNDIVIDED box across every seam and the wrap (window.GlobalCheck): each rank hands in its
        # share of the windows, rank 0 merges and advances them with no notion of slabs
        def gather(obj):
            # all_gather_object: served by RCCL process groups and by gloo alike (gather_object is not, on every torch)
            got = [None] * world
            torch.distributed.all_gather_object(got, obj)
            return got if rank == 0 else None
        gz = nz * world if axis == 'z' else nz
        glob = window.GlobalCheck(backend, sim.desc, sorted(set([1, gz // 2 + 1])), sim.gpu_dist, sim.stride, AXES[axis],
                                  rank, world, gather)
        glob.seed(sim.iteration)
    chk.seed(sim.iteration)
    sim.step()
    sim.step(save_macro=True)
    sim.sync()
    if sim.halo:
        sim.materialise_faces()
    chk.advance(2, save_last=True)
    r = chk.compare()
    if glob is not None:
        glob.advance(2)
        g = glob.compare()
        if rank == 0:
            out['undivided_box'] = {'box': g['box'], 'slabs': g['slabs'], 'planes': g['planes'],
                                    'populations_compared': g['compared_values'],
                                    'populations_bit_identical': g['dist_exact'], 'max_abs_err': g['dist_err'],
                                    'what': 'whole planes of the merged slabs against oracle windows of the undivided '
                                            'periodic box (window.GlobalCheck)'}
    out.update(planes=r['planes'], populations_compared=r['compared_values'], populations_bit_identical=r['dist_exact'],
               max_abs_err=r['dist_err'], rho_rel_err=r['rho_err'], u_abs_err=r['v_abs_err'])
    rho, v = sim.fetch_fields()
    r64 = sim.real_view(rho).astype(np.float64)
    tot = [float(r64.sum())] + [float((r64 * sim.real_view(c)).sum()) for c in v]
    if distributed:
        t = torch.tensor(tot + list(mass0), dtype=torch.float64, device='cuda')
        torch.distributed.all_reduce(t)
        tot, mass0 = t[:4].tolist(), t[4:].tolist()
    out['mass_rel_drift'] = abs(tot[0] - mass0[0]) / mass0[0]
    out['momentum_drift_over_mass_u'] = max(abs(a - b) for a, b in zip(tot[1:], mass0[1:])) / (mass0[0] * 0.05)
    out['steps_before_check'] = sim.iteration - 2
    # f32 round-off random-walks the totals: 8e-6 after 540 steps at 512^3, 1.6e-5 after 1500; a lost face layer is 2e-3
    out['ok'] = bool(out['mass_rel_drift'] < 5e-5 and out['momentum_drift_over_mass_u'] < 2e-4 and
                     out.get('populations_bit_identical', True) and out.get('rho_rel_err', 0.0) < 1e-6 and
                     out.get('undivided_box', {}).get('populations_bit_identical', True))
    if distributed:                      # every rank checked its own planes: all of them must agree
        flags = [1.0 if out['ok'] else 0.0, 1.0 if out.get('populations_bit_identical', True) else 0.0]
        t = torch.tensor(flags, dtype=torch.float64, device='cuda')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        out['ok'] = bool(t[0].item() > 0.5)
        out['populations_bit_identical'] = bool(t[1].item() > 0.5)
        out['ranks_checked'] = world
    if os.environ.get('SLF_BENCH_TEST_REJECT_PEER') == '1' and getattr(sim, 'peer', None) is not None:
        out['ok'] = False           # tests: what the line looks like when the peer transport's results are not accepted
        out['test_forced_rejection'] = True
    return out


def device_report(local_rank):
    """What this rank runs on and which other devices of the node it can reach directly (RCCL's xGMI paths)."""
    import torch
    rep = {'device': local_rank}
    try:
        p = torch.cuda.get_device_properties(local_rank)
        rep['name'] = p.name
        rep['uuid'] = str(getattr(p, 'uuid', ''))
        rep['pci_bus_id'] = '%04x:%02x:%02x.0' % (getattr(p, 'pci_domain_id', 0), getattr(p, 'pci_bus_id', 0), getattr(p, 'pci_device_id', 0))
        n = torch.cuda.device_count()
        rep['peer_access'] = [int(i == local_rank or torch.cuda.can_device_access_peer(local_rank, i)) for i in range(n)]
    except Exception as e:  # noqa: BLE001
        rep['error'] = str(e)[:100]
    return rep


class NoWatchdog(object):
    """Single process, no neighbours: nothing to wait for."""
    def phase(self, *a, **k):
        pass
    note = failed = close = phase


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if args.gpus > 1 and world == 1:
        raise SystemExit(self_launch(args.gpus))
    wd = NoWatchdog()
    if world > 1 or args.force_distributed:
        # every phase of a multi-process run has a deadline (sailfish_amd/watchdog.py): a rendezvous, a communicator or
        # an exchange that never completes ends in ONE line with "error", the phase and every rank's last state, and a
        # non-zero status -- not in the launcher's time limit (reference: master.py:268-312 polls and tears down)
        from sailfish_amd.watchdog import Watchdog

        def report(diag):
            out = {'metric': 'MLUPS (million lattice updates/s), D3Q19 %s' % args.model.upper(), 'value': None, 'unit': 'MLUPS',
                   'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': True, 'scaling': args.scaling}
            out.update(diag)
            print(json.dumps(out), flush=True)
        wd = Watchdog(rank, world, report, extra=lambda: WATCH_DETAIL() if WATCH_DETAIL else None)
    try:
        run(args, world, rank, wd)
    except BaseException as e:      # noqa: BLE001 -- the others must hear of it before this process goes
        if not isinstance(e, SystemExit) or e.code not in (0, None):
            wd.failed('%s: %s' % (type(e).__name__, e))
            if rank == 0 and not isinstance(wd, NoWatchdog):
                time.sleep(0.5)
                wd._fire('rank 0: %s: %s' % (type(e).__name__, str(e)[:300]), 'failed')
        raise
    wd.close()


WATCH_DETAIL = None      # set by run(): what the watchdog adds to a rank's state (transport counters)


def run(args, world, rank, wd):
    global WATCH_DETAIL
    import torch
    # SLF_FORCE_DEVICE: every rank on that GPU (functional runs of the N > 1 path on a 1-GPU box, with SLF_DIST_BACKEND=gloo)
    local_rank = int(os.environ.get('SLF_FORCE_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP backend has no CPU fallback)')
    torch.cuda.set_device(local_rank)

    from sailfish_amd import sym
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.slab import AXES, SlabSim

    class Opt(object):
        pass

    backend = HIPBackend(Opt(), local_rank)
    distributed = world > 1 or args.force_distributed
    pinned = None
    if world > 1 and os.environ.get('SLF_PIN_RANKS', '1') != '0':
        # every rank on cores of its GPU's NUMA node, the node's cores dealt out between the ranks that share it (what
        # the controller does for the subdomain processes it starts: sailfish_amd/launch.py)
        try:
            from sailfish_amd import launch
            per_node = int(os.environ.get('LOCAL_WORLD_SIZE', world))
            me = int(os.environ.get('LOCAL_RANK', '0')) % per_node
            forced = os.environ.get('SLF_FORCE_DEVICE')
            nodes = [launch.gpu_numa_node(HIPBackend.pci_bus_id(int(forced) if forced is not None else d))
                     for d in range(per_node)]
            sets = launch.cpu_sets(nodes, sorted(os.sched_getaffinity(0)), launch.numa_cpus)
            os.sched_setaffinity(0, sets[me])
            pinned = {'numa_node': nodes[me], 'cpus': len(sets[me])}
        except Exception as e:  # noqa: BLE001 -- pinning is an optimisation, never a reason to fail
            pinned = {'error': str(e)[:80]}
    test_stall(wd, rank, 'start')
    if distributed:
        from sailfish_amd.connector import init_distributed, process_rccl
        from sailfish_amd import peer as peer_mod
        wd.phase('rendezvous')
        init_distributed(force=True)
        test_stall(wd, rank, 'rendezvous')
        # the transport of the halos, set up here so that it has a phase (and a deadline) of its own: peer mappings where
        # the ranks can map each other's memory, else an RCCL communicator (ncclCommInitRank is where a multi-GPU run
        # that cannot reach its peers stops)
        wd.phase('transport')
        pt = peer_mod.process_transport(backend, rank, world)
        if pt is not None:
            WATCH_DETAIL = pt.snapshot
        elif torch.distributed.get_backend() == 'nccl' and os.environ.get('SLF_HALO_TRANSPORT', 'auto') != 'torch':
            process_rccl(backend, rank, world)
        test_stall(wd, rank, 'transport')
    axis = AXES[args.axis]
    if args.scaling == 'weak':
        local = [args.size] * 3
        domain = list(local)
        domain[axis] *= world
    else:
        domain = [int(x) for x in args.domain.split('x')]
        if domain[axis] % world:
            raise SystemExit('the %s extent of --domain must be a multiple of the rank count' % args.axis)
        local = list(domain)
        local[axis] //= world
    local_nodes = local[0] * local[1] * local[2]

    def barrier(sim):
        sim.sync()
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()

    def measure(pattern, check=False):
        wd.phase('setup', pattern=pattern)
        sim = SlabSim(backend, sym.D3Q19, tuple(local), rank=rank, world=world, model=args.model,
                      precision=args.precision, access_pattern=pattern, visc=args.visc,
                      fused_periodic=not args.no_fused_periodic, axis=args.axis, force_halo=args.force_distributed,
                      tune_placement=not args.no_placement_tune)
        res = {'pattern': pattern, 'block': sim.block_size, 'placement': sim.placement_info}
        if sim.placement_tuning:
            res['placement'] = dict(sim.placement_info or {}, tuning=sim.placement_tuning)
        sim.init_synthetic(seed=1234)
        if check:
            r64 = sim.real_view(sim.rho).astype(np.float64)
            mass0[0] = [float(r64.sum())] + [float((r64 * sim.real_view(c)).sum()) for c in sim.v]
        if sim.halo:
            wd.phase('first_exchange', pattern=pattern)
            for _ in range(2):
                sim.step()
            barrier(sim)
            test_stall(wd, rank, 'first_exchange')
            wd.phase('setup', pattern=pattern, what='sweep-only reference')
            # what the sweep launches cost when nothing is waited for (reference for the overlap figure)
            for _ in range(10):
                sim.step_sweep_only()
            sim.sync()
            e0 = backend.make_event(sim.calc_stream, timing=True)
            for _ in range(20):
                sim.step_sweep_only()
            e1 = backend.make_event(sim.calc_stream, timing=True)
            e1.synchronize()
            res['sweep_only_ms'] = e1.time_since(e0) / 20
            sim.init_synthetic(seed=1234)
        wd.phase('warmup', pattern=pattern)
        for _ in range(args.prewarm_steps + (args.prewarm_steps & 1)):   # untimed, even count: GPU clocks ramp up
            sim.step()
        for _ in range(args.warmup):
            sim.step()
        test_stall(wd, rank, 'warmup')
        # timed region: blocks of EXACTLY K steps, each bracketed by barrier + synchronize, max over ranks; repeated until
        # --min_seconds have been timed (the driver's K = 20 is 64 ms at 512^3: too short a sample on its own)
        blocks, host, kernel = [], [], []
        total = 0.0
        while True:
            wd.phase('timed', pattern=pattern, block=len(blocks))
            barrier(sim)
            ev0 = backend.make_event(sim.calc_stream, timing=True)
            t0 = time.perf_counter()
            per_step = []
            for _ in range(args.steps):
                t1 = time.perf_counter()
                sim.step()
                per_step.append(time.perf_counter() - t1)
            ev1 = backend.make_event(sim.calc_stream, timing=True)
            barrier(sim)
            elapsed = time.perf_counter() - t0
            if distributed:
                t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                elapsed = float(t.item())
            ev1.synchronize()
            blocks.append(elapsed)
            kernel.append(ev1.time_since(ev0) / args.steps)   # HIP events on the sweep's own stream
            host.append(per_step)
            total += elapsed
            if total >= args.min_seconds or len(blocks) >= 200:
                break
        res['elapsed'] = blocks[median_block(blocks)]    # the median block; the best one is reported beside it
        res['blocks'] = blocks
        res['kernels'] = kernel                          # per block: HIP events on the sweep's own stream / K
        res['kernel_ms'] = kernel[median_block(blocks)]
        # enqueueing time per step.  mean: includes the stretches in which the runtime / RCCL made the host wait because
        # their queues were full (the host runs far ahead of the GPU: that wait is harmless); median: what a step costs
        # the host when nothing holds it back
        flat = np.array([x for b in host for x in b]) * 1e3
        res['host_ms'] = float(flat.mean())
        res['host_ms_median'] = float(np.median(flat))
        if sim.halo:
            # a separate short leg with timing events around the halo stream's work (entry-by-entry enqueue: the timed
            # region above replays step plans, which carry no timing events)
            wd.phase('halo_timing', pattern=pattern)
            sim.start_halo_timing()
            for _ in range(args.halo_timing_steps + (args.halo_timing_steps & 1)):
                sim.step()
            res['halo_ms'] = sim.stop_halo_timing()
            res['step_plans'] = sorted(str(k) for k in getattr(sim, '_plans', {}))
            res['rccl_ranks'] = rccl_ranks_of(getattr(sim, 'exchanger', None),
                                              torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                                              world, getattr(sim, 'peer', None))
            res['halo_transport'] = halo_transport_of(sim)
        if check:
            wd.phase('validate', pattern=pattern)
            res['validation'] = validate(sim, backend, mass0[0], distributed, args.axis, rank, world)
        wd.phase('finish', pattern=pattern)
        sim.release()
        return res

    st_before = None if (args.no_gpu_state or rank) else gpu_state()
    patterns = ['AA', 'AB'] if args.access_pattern == 'auto' else [args.access_pattern]
    runs = dict((p, []) for p in patterns)
    mass0 = [None]
    reps = max(1, args.repeats)
    for i in range(reps):
        for pat in patterns:
            runs[pat].append(measure(pat, check=(i == reps - 1 and not args.no_validate)))
    rejected = None
    if distributed and not args.no_validate and os.environ.get('SLF_HALO_TRANSPORT', 'auto') == 'auto':
        # the peer transport has only ever run between processes that SHARE a device (the build pool has single-GPU boxes):
        # if its start-up check passed on this node but the seam layers of the final state do not match the oracle, the
        # figures above are not reported -- the run is repeated over RCCL / torch.distributed and the line says so.  (The
        # verdicts were combined over the ranks inside validate(): every rank takes the same branch.)
        bad = dict((p, rs[-1]['validation']) for p, rs in runs.items()
                   if 'validation' in rs[-1] and not rs[-1]['validation']['ok'] and (rs[-1].get('halo_transport') or '').startswith('peer'))
        if bad:
            rejected = {'halo_transport': 'peer', 'mlups': dict((p, round(local_nodes * world * args.steps / runs[p][-1]['elapsed'] * 1e-6, 1)) for p in bad),
                        'validation': bad, 'what': 'seam layers written through the peer mappings did not match the oracle; repeated '
                                                    'over RCCL / torch.distributed, whose figures this line reports'}
            os.environ['SLF_HALO_TRANSPORT'] = 'rccl' if torch.distributed.get_backend() == 'nccl' else 'torch'
            runs = dict((p, []) for p in patterns)
            for i in range(reps):
                for pat in patterns:
                    runs[pat].append(measure(pat, check=(i == reps - 1)))
    st_after = None if (args.no_gpu_state or rank) else gpu_state()
    wd.phase('report')
    # per access pattern: every timed block of every repeat; the pattern with the better MEDIAN block is reported
    pooled = dict((p, [b for r in rs for b in r['blocks']]) for p, rs in runs.items())
    med_of = dict((p, bl[median_block(bl)]) for p, bl in pooled.items())
    args.access_pattern = min(med_of, key=med_of.get)
    elapsed = med_of[args.access_pattern]
    # the sweep time of that very block
    kernel_ms = [k for r in runs[args.access_pattern] for k in r['kernels']][median_block(pooled[args.access_pattern])]
    best = min(runs[args.access_pattern], key=lambda r: r['elapsed'])
    per_rank = None
    if distributed:
        mine = dict((k, round(best[k], 4)) for k in ('kernel_ms', 'halo_ms', 'sweep_only_ms', 'host_ms', 'host_ms_median') if k in best)
        mine['rank'] = rank
        mine['step_plans'] = best.get('step_plans')
        mine['halo_transport'] = best.get('halo_transport')
        mine['pinned'] = pinned
        mine.update(device_report(local_rank))
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, mine)
        per_rank = gathered

    total_nodes = local_nodes * world
    to_mlups = lambda t: total_nodes * args.steps / t * 1e-6   # noqa: E731
    prec = 4 if args.precision == 'single' else 8
    if rank == 0:
        bpu = BYTES_PER_UPDATE[prec]
        achieved = local_nodes * bpu / (kernel_ms * 1e-3) / 1e9
        shape = 'x'.join(str(v) for v in local)
        wkey = 'D3Q19_%s_f%d_%s_%s_%s' % (args.model, prec * 8, args.access_pattern,
                                          str(local[0]) if len(set(local)) == 1 else shape,
                                          'ghostpbc' if args.no_fused_periodic else 'fused')
        all_mlups = sorted(to_mlups(b) for r in runs[args.access_pattern] for b in r['blocks'])
        if args.scaling == 'weak':
            what = 'D3Q19 %s %s^3 per GPU' % (args.model.upper(), args.size)
        else:
            what = 'D3Q19 %s %s box over %d GPU(s)' % (args.model.upper(), 'x'.join(map(str, domain)), world)
        cfg = {'workload': 'D3Q19 %s periodic box, %s nodes per GPU, global %s (fluid nodes only counted)'
                           % (args.model.upper(), shape, 'x'.join(map(str, domain))),
               'access_pattern': args.access_pattern,
               'periodic': 'in-sweep wrap' if not args.no_fused_periodic else 'ghost-layer PBC kernels',
               'decomposition': ('%s-slabs x%d, halo: %s' % (args.axis, world, (best.get('halo_transport') or '?').split(':')[0].split(' (')[0]))
               if distributed else 'single subdomain',
               'visc': args.visc, 'block_x': best['block'], 'repeats': max(1, args.repeats),
               'value_is': 'median block of exactly K steps (blocks repeated until --min_seconds are timed, per repeat); '
                           'best_value = the fastest block',
               'timed_blocks': len(all_mlups), 'timed_seconds': round(sum(b for r in runs[args.access_pattern] for b in r['blocks']), 3),
               'median_mlups': round(to_mlups(elapsed), 1),
               'runs_mlups': [round(v, 1) for v in (all_mlups if len(all_mlups) <= 12 else all_mlups[:4] + all_mlups[-8:])],
               'candidates_mlups': dict((p, round(to_mlups(t), 1)) for p, t in med_of.items()),
               'placement': best['placement']}
        checks = dict((p, rs[-1]['validation']) for p, rs in runs.items() if 'validation' in rs[-1])
        if checks:
            cfg['validated'] = all(c['ok'] for c in checks.values())
            cfg['validation'] = checks
        if rejected:
            cfg['peer_transport_rejected'] = rejected
        if distributed:
            hm = max(r.get('halo_ms', 0.0) for r in per_rank)
            so = max(r.get('sweep_only_ms', 0.0) for r in per_rank)
            step_ms = elapsed / args.steps * 1e3
            exposed = max(0.0, step_ms - so)
            cfg.update({'rccl_ranks': best.get('rccl_ranks', 0),
                        'rccl_ranks_is': 'ncclCommCount of the communicator the halos travel over (C ABI slf_comm_count); '
                                         '0 = not over RCCL',
                        'world_size': torch.distributed.get_world_size(),
                        'dist_backend': torch.distributed.get_backend(),
                        'halo_transport': best.get('halo_transport'),
                        'host_ms_median': round(max(r.get('host_ms_median', 0.0) for r in per_rank), 4),
                        'per_rank': per_rank,
                        'halo_overlap_frac': round(max(0.0, min(1.0, 1.0 - exposed / hm)), 3) if hm > 0 else None,
                        'halo_exposed_ms': round(exposed, 4)})
        if st_before or st_after:
            cfg['gpu_state'] = {'before': st_before, 'after': st_after}
        out = {
            'metric': 'MLUPS (million lattice updates/s), %s' % what,
            'value': round(to_mlups(elapsed), 1), 'value_kind': 'median block (rounds 3-4 reported the fastest block: best_value)',
            'best_value': round(max(all_mlups), 1),
            'unit': 'MLUPS', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32' if prec == 4 else 'f64', 'data': 'synthetic',
            'config': cfg,
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': load_traffic(wkey),
                         'bytes_per_update': bpu, 'kernel_ms': round(kernel_ms, 4)},
        }
        if world == 1 and not args.no_runner_path and not distributed and args.scaling == 'weak':
            cfg['runner_path'] = runner_path(args, local_rank)
            cfg['runner_path']['vs_value'] = round(cfg['runner_path']['mlups'] / out['value'], 4)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(out))
    if distributed:
        wd.phase('finish')
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def test_stall(wd, rank, phase):
    """SLF_BENCH_TEST_STALL=<rank>:<phase>[:exit]: that rank stops (or leaves) at the end of that phase -- what the tests of
    the deadlines use to make a run hang the way a lost neighbour would."""
    spec = os.environ.get('SLF_BENCH_TEST_STALL')
    if not spec:
        return
    parts = spec.split(':')
    if int(parts[0]) != rank or parts[1] != phase:
        return
    if len(parts) > 2 and parts[2] == 'exit':
        os._exit(17)
    while True:
        time.sleep(1.0)


if __name__ == '__main__':
    main()
