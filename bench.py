#!/usr/bin/env python3
"""Headline benchmark: MLUPS of the fused collide-and-stream sweep, D3Q19 BGK,
single precision, 512^3 fully periodic box per GPU (BASELINE.json metric;
SURVEY.md §8(d) "M0"), through backend_hip -> libsailfish_hip.so.

    python bench.py --gpus N --steps K --warmup W

N > 1: launched by torch.distributed.run, one rank per GPU; the global domain
is 512 x 512 x (512 N), cut into N slabs along z (weak scaling), halo planes
exchanged device-to-device over RCCL every step (sailfish_amd/connector.py).

Prints ONE JSON line on rank 0 (see the field list in DESIGN.md §6).
"""
import argparse
import json
import os
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
    ap.add_argument('--size', type=int, default=512, help='box edge per GPU (512 = the headline config)')
    ap.add_argument('--access_pattern', default='auto', choices=['auto', 'AA', 'AB'],
                    help='AA = in-place single copy, AB = two copies (the reference default); auto times both '
                         'with the same K steps and reports the faster one')
    ap.add_argument('--model', default='bgk', choices=['bgk', 'mrt'])
    ap.add_argument('--precision', default='single', choices=['single', 'double'])
    ap.add_argument('--no_fused_periodic', action='store_true',
                    help='use ghost-layer PBC kernels (reference scheme) instead of in-sweep wrap')
    ap.add_argument('--visc', type=float, default=1.0 / 6.0)
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--cpu_seconds', type=float, default=12.0)
    ap.add_argument('--repeats', type=int, default=2,
                    help='each candidate access pattern is timed this many times (K steps each); the best is kept')
    ap.add_argument('--prewarm_steps', type=int, default=300,
                    help='untimed steps before the W warm-up steps (same count on every rank), so that clocks and '
                         'caches are in steady state: about 1 s at 512^3')
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (CPU restatement, OpenMP) timed on this box's host cores on a bounded sample of
    the same workload: D3Q19 BGK AA periodic box, 128^3, as many steps as fit in ~cpu_seconds.
    The reference has no CPU compute path, hence kind = "port"."""
    from sailfish_amd import sym
    from sailfish_amd.box import make_box_desc
    from tests._oracle_box import OracleBox, synthetic_fields
    n = 128
    size = (n, n, n)
    desc = make_box_desc(sym.D3Q19, size, model=args.model, precision=args.precision, access_pattern='AA',
                         visc=args.visc, periodic_fused=[1, 1, 1])
    ob = OracleBox(desc, periodic=(True, True, True))
    rho, v = synthetic_fields(size, 3, dtype=np.float32)
    ob.set_fields(rho, v)
    ob.initial_conditions()
    ob.run(2, save_last=False)
    t0 = time.time()
    steps = 0
    while time.time() - t0 < args.cpu_seconds:
        ob.run(2, save_last=False)
        steps += 2
    dt = time.time() - t0
    cores = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
    cpu_model = 'unknown CPU'
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    cpu_model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'value': round(n ** 3 * steps / dt * 1e-6, 2), 'unit': 'MLUPS', 'cores': cores, 'kind': 'port',
            'sample': 'oracle/lbm_oracle.c (OpenMP, %d threads on %s), D3Q19 %s f%d AA periodic %d^3, %d steps in %.1f s'
                      % (cores, cpu_model, args.model.upper(), 32 if args.precision == 'single' else 64, n, steps, dt)}


def load_traffic(workload_key):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC runs (profiles/traffic.json),
    if one exists for this exact workload."""
    p = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(p) as fh:
            return json.load(fh).get(workload_key)
    except Exception:
        return None


def main():
    args = parse_args()
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit('N > 1 must be launched with torch.distributed.run (one rank per GPU)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP backend has no CPU fallback)')
    torch.cuda.set_device(local_rank)

    from sailfish_amd import sym
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.slab import SlabSim

    class Opt(object):
        pass

    backend = HIPBackend(Opt(), local_rank)
    n = args.size

    def barrier(sim):
        sim.sync()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    def measure(pattern):
        sim = SlabSim(backend, sym.D3Q19, (n, n, n), rank=rank, world=world, model=args.model,
                      precision=args.precision, access_pattern=pattern, visc=args.visc,
                      fused_periodic=not args.no_fused_periodic)
        sim.init_synthetic(seed=1234)
        for _ in range(args.prewarm_steps + (args.prewarm_steps & 1)):   # untimed, even count: GPU clocks ramp up
            sim.step()
        for _ in range(args.warmup):
            sim.step()
        barrier(sim)
        ev0 = backend.make_event(sim.calc_stream, timing=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sim.step()
        ev1 = backend.make_event(sim.calc_stream, timing=True)
        barrier(sim)
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        ev1.synchronize()
        kernel_ms = ev1.time_since(ev0) / args.steps   # HIP events on the sweep's own stream
        res = {'pattern': pattern, 'elapsed': elapsed, 'kernel_ms': kernel_ms, 'block': sim.block_size}
        sim.release()
        return res

    patterns = ['AA', 'AB'] if args.access_pattern == 'auto' else [args.access_pattern]
    results = {}
    for _ in range(max(1, args.repeats)):
        for pat in patterns:
            r = measure(pat)
            if pat not in results or r['elapsed'] < results[pat]['elapsed']:
                results[pat] = r
    results = [results[p] for p in patterns]
    best = min(results, key=lambda r: r['elapsed'])
    elapsed, kernel_ms = best['elapsed'], best['kernel_ms']
    args.access_pattern = best['pattern']

    fluid_nodes = n ** 3 * world
    mlups = fluid_nodes * args.steps / elapsed * 1e-6
    prec = 4 if args.precision == 'single' else 8
    if rank == 0:
        bpu = BYTES_PER_UPDATE[prec] if True else None
        achieved = n ** 3 * bpu / (kernel_ms * 1e-3) / 1e9
        wkey = 'D3Q19_%s_f%d_%s_%d_%s' % (args.model, prec * 8, args.access_pattern, n,
                                          'ghostpbc' if args.no_fused_periodic else 'fused')
        out = {
            'metric': 'MLUPS (million lattice updates/s), D3Q19 BGK 512^3',
            'value': round(mlups, 1), 'unit': 'MLUPS', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if prec == 4 else 'f64', 'data': 'synthetic',
            'config': {'workload': 'D3Q19 %s periodic box %d^3 per GPU (fluid nodes only counted)'
                                   % (args.model.upper(), n),
                       'access_pattern': args.access_pattern,
                       'periodic': 'in-sweep wrap' if not args.no_fused_periodic else 'ghost-layer PBC kernels',
                       'decomposition': 'z-slabs x%d, RCCL halo' % world if world > 1 else 'single subdomain',
                       'visc': args.visc, 'block_x': best['block'], 'repeats': max(1, args.repeats),
                       'candidates_mlups': dict((r['pattern'], round(fluid_nodes * args.steps / r['elapsed'] * 1e-6, 1))
                                                for r in results)},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': load_traffic(wkey),
                         'bytes_per_update': bpu, 'kernel_ms': round(kernel_ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
