#!/usr/bin/env python3
"""MLUPS of every BASELINE.json configuration that fits one GPU, through the full host stack
(examples -> LBSimulationController --mode=benchmark -> backend_hip), one JSON line per configuration.
bench.py stays the headline number (config: 512^3 periodic box); this is the table behind DESIGN.md.

    python tools/bench_configs.py [--out profiles/r01/configs.jsonl] [--quick]
"""
import argparse
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)

from sailfish.controller import LBSimulationController  # noqa: E402
from sailfish.geo import EqualSubdomainsGeometry3D, LBGeometry2D, LBGeometry3D  # noqa: E402
from sailfish.lb_single import LBFluidSim  # noqa: E402
from sailfish.subdomain import Subdomain3D  # noqa: E402

BYTES = {('D3Q19', 'single'): 152, ('D3Q19', 'double'): 304, ('D2Q9', 'single'): 72}


class PeriodicBox(Subdomain3D):
    def boundary_conditions(self, hx, hy, hz):
        pass

    def initial_conditions(self, sim, hx, hy, hz):
        import numpy as np
        sim.rho[:] = 1.0
        sim.vx[:] = 0.05 * np.sin(2 * np.pi * hy / self.gy)
        sim.vy[:] = 0.05 * np.sin(2 * np.pi * hz / self.gz)
        sim.vz[:] = 0.05 * np.sin(2 * np.pi * hx / self.gx)


class BoxSim(LBFluidSim):
    subdomain = PeriodicBox


class OpenDuct(Subdomain3D):
    """A duct with an inlet and an outlet (not a BASELINE configuration; ids 6*): full-way bounce-back walls on the two y
    faces, a regularized-velocity inlet and an NTDoNothing outlet on the two faces of `flow_axis`, periodic along the
    third axis.  Along x every row holds an inlet and an outlet node, so every row is swept by the instantiation with
    the boundary-condition code; along z only the rows of the first and the last plane are."""
    flow_axis = 0
    u_in = 0.03

    def boundary_conditions(self, hx, hy, hz):
        from sailfish.node_type import NTDoNothing, NTFullBBWall, NTRegularizedVelocity
        wall = (hy == 0) | (hy == self.gy - 1)
        h, g = ((hx, self.gx), None, (hz, self.gz))[self.flow_axis]
        v = [0.0, 0.0, 0.0]
        v[self.flow_axis] = self.u_in
        self.set_node(wall, NTFullBBWall)
        self.set_node(~wall & (h == 0), NTRegularizedVelocity(tuple(v)))
        self.set_node(~wall & (h == g - 1), NTDoNothing)

    def initial_conditions(self, sim, hx, hy, hz):
        sim.rho[:] = 1.0
        (sim.vx, None, sim.vz)[self.flow_axis][:] = self.u_in


class OpenDuctZ(OpenDuct):
    flow_axis = 2


class DuctSimX(LBFluidSim):
    subdomain = OpenDuct


class DuctSimZ(LBFluidSim):
    subdomain = OpenDuctZ


def porous_wall_map(nx, ny, nz, fluid_fraction=0.3, radius=12, seed=7):
    """Boolean [nz, ny, nx] wall map of a packed bed: overlapping solid spheres of one radius at random positions
    (periodic images included), as many as leave about `fluid_fraction` of the box to the fluid."""
    import numpy as np
    rng = np.random.RandomState(seed)
    vol = 4.0 / 3.0 * np.pi * radius ** 3
    n = int(-np.log(fluid_fraction) * nx * ny * nz / vol)
    wall = np.zeros((nz, ny, nx), dtype=bool)
    r = int(radius) + 1
    d = np.arange(-r, r + 1)
    ball = (d[:, None, None] ** 2 + d[None, :, None] ** 2 + d[None, None, :] ** 2) <= radius ** 2
    cz, cy, cx = rng.randint(0, nz, n), rng.randint(0, ny, n), rng.randint(0, nx, n)
    for k in range(n):
        zi, yi, xi = (cz[k] + d) % nz, (cy[k] + d) % ny, (cx[k] + d) % nx
        wall[np.ix_(zi, yi, xi)] |= ball
    return wall


def _run(label, sim_cls, geo, settings, bytes_per_update):
    cfg = dict(mode='benchmark', quiet=True, perf_stats_every=0)
    cfg.update(settings)
    ctrl = LBSimulationController(sim_cls, geo, default_config=cfg)
    with redirect_stdout(io.StringIO()):
        ctrl.run(ignore_cmdline=True)
    nodes = sum(r.num_fluid_nodes for r in ctrl.runners)
    out = {'config': label, 'MLUPS_eff': round(ctrl.mlups_total, 1), 'MLUPS_comp': round(ctrl.mlups_comp, 1),
           'fluid_nodes': int(nodes), 'settings': dict((k, v) for k, v in settings.items() if not k.startswith('_'))}
    shared = len(ctrl.runners) > 1 and len(set(r.backend.gpu_id for r in ctrl.runners)) < len(ctrl.runners)
    r0 = ctrl.runners[0]
    dense = 1
    for n in r0._lat_size:
        dense *= max(1, n - 2)
    out['fluid_fraction'] = round(float(nodes) / (dense * len(ctrl.runners)), 4) if len(ctrl.runners) == 1 else None
    if getattr(r0, 'indirect', False):
        out['active_slots'] = int(r0._dist_stride)
    if os.environ.get('SLF_BENCH_DETAIL'):       # per-subdomain mean / fastest / slowest minibatch (ms per step)
        out['step_ms'] = [dict(id=r._spec.id, mean=round(r.summary[0].total * 1e3, 4), min=round(r.summary[1].total * 1e3, 4),
                               max=round(r.summary[2].total * 1e3, 4), comp=round(r.summary[0].comp * 1e3, 4),
                               coll=round(r.summary[0].coll * 1e3, 4)) for r in ctrl.runners if r.summary]
    tun = [getattr(r, 'placement_tuning', None) for r in ctrl.runners]
    if any(tun):
        out['placement_tuning'] = tun
    if callable(bytes_per_update):
        bytes_per_update = bytes_per_update(out)
    if bytes_per_update:
        out['bytes_per_update'] = round(bytes_per_update, 1)
        out['GBps_eff'] = round(ctrl.mlups_total * bytes_per_update / 1e3, 1)   # wall clock, everything included
        if shared:
            # several subdomains TIME-SHARE one device: the reference's `comp` -- the sum over subdomains of nodes / own
            # kernel time (controller.py:471-476) -- adds up rates that were never achieved at the same time; the
            # wall-clock figure is the only one that means anything
            out['MLUPS_comp'] = None
            out['frac_of_8TBps'] = round(ctrl.mlups_total * bytes_per_update / 8e6, 4)
            out['frac_is'] = 'wall clock (subdomains share the device: no per-kernel rate)'
        else:
            out['GBps_comp'] = round(ctrl.mlups_comp * bytes_per_update / 1e3, 1)   # sweep kernels only
            out['frac_of_8TBps'] = round(ctrl.mlups_comp * bytes_per_update / 8e6, 4)
    print(json.dumps(out), flush=True)
    for r in ctrl.runners:      # give the device memory back before the next configuration
        r.release()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--only', default='', help='comma separated configuration ids (0,1,2,2b,2c,5a,5b,3,3b,3g8,3x8,3z8,4; not in the default set: 6xa,6xb,6za,6zb, 7a-7d indirect addressing, 8a,8b single-component Shan-Chen, 9a,9b cylinder / sphere, 4x4,8x4,5x3 x-slabs of one process)')
    args = ap.parse_args()
    if not args.only:
        # one fresh process per configuration: what a configuration measures must not depend on what ran before it in
        # the same process (stream -> hardware-queue mapping, allocator state; profiles/r02/README.md "config 3")
        import subprocess
        lines = []
        for cid in ('0', '1', '2', '2b', '2c', '5a', '5b', '3', '3b', '3g8', '3x8', '3z8', '4'):
            cmd = [sys.executable, os.path.abspath(__file__), '--only', cid] + (['--quick'] if args.quick else [])
            out = subprocess.run(cmd, stdout=subprocess.PIPE).stdout.decode(errors='replace')
            for ln in out.splitlines():
                if ln.startswith('{'):
                    print(ln, flush=True)
                    lines.append(ln)
        if args.out:
            with open(args.out, 'w') as fh:
                fh.write('\n'.join(lines) + '\n')
        return
    from examples.ldc_2d import CavitySim as Cavity2D
    from examples.ldc_3d import CavitySim as Cavity3D
    from examples.binary_fluid.sc_separation_3d import SeparationSim
    it = 0.3 if args.quick else 1.0
    only = set(x for x in args.only.split(',') if x)

    res = []

    def run(label, *a, **kw):       # noqa: F811 -- filters by id, then the module-level run()
        if only and label.split(':')[0] not in only:
            return None
        return _run(label, *a, **kw)
    # config 0: the reference's own small case, on the GPU here (its CPU figure is bench.py's cpu_baseline)
    res.append(run('0: ldc_2d D2Q9 BGK 256x256', Cavity2D, LBGeometry2D,
                   dict(lat_nx=256, lat_ny=256, visc=0.0254, access_pattern='AA', max_iters=int(40000 * it),
                        benchmark_sample_from=int(10000 * it)), 72))
    # config 1: D3Q19 BGK periodic box 256^3
    res.append(run('1: D3Q19 BGK periodic box 256^3', BoxSim, LBGeometry3D,
                   dict(lat_nx=256, lat_ny=256, lat_nz=256, periodic_x=True, periodic_y=True, periodic_z=True,
                        visc=1.0 / 6.0, access_pattern='AA', grid='D3Q19', max_iters=int(3000 * it),
                        benchmark_sample_from=int(1000 * it)), 152))
    # config 2: D3Q19 MRT lid-driven cavity 512^3 (Re = 1000)
    res.append(run('2: D3Q19 MRT lid-driven cavity 512^3', Cavity3D, LBGeometry3D,
                   dict(lat_nx=512, lat_ny=512, lat_nz=512, model='mrt', visc=0.0256, access_pattern='AA',
                        max_iters=int(700 * it), benchmark_sample_from=int(200 * it)), 152))
    res.append(run('2b: D3Q19 BGK lid-driven cavity 512^3', Cavity3D, LBGeometry3D,
                   dict(lat_nx=512, lat_ny=512, lat_nz=512, model='bgk', visc=0.0256, access_pattern='AA',
                        max_iters=int(700 * it), benchmark_sample_from=int(200 * it)), 152))
    res.append(run('2c: D3Q19 BGK lid-driven cavity 512^3, two-copy (AB) pattern = the reference default', Cavity3D,
                   LBGeometry3D,
                   dict(lat_nx=512, lat_ny=512, lat_nz=512, model='bgk', visc=0.0256, access_pattern='AB',
                        max_iters=int(700 * it), benchmark_sample_from=int(200 * it)), 152))
    # a force-driven pipe (examples/poiseuille_3d.py: circular cross-section of full-way bounce-back walls, unused nodes
    # outside, periodic along x, body force): the node-map kernels with the Guo term, in-place and two-copy
    from examples.poiseuille_3d import PipeSim as Pipe3D
    for cid, pattern in (('5a', 'AA'), ('5b', 'AB')):
        res.append(run('%s: D3Q19 BGK force-driven pipe 512x256x256 (%s)' % (cid, pattern), Pipe3D, LBGeometry3D,
                       dict(lat_nx=512, lat_ny=256, lat_nz=256, visc=0.05, access_pattern=pattern,
                            max_iters=int(1500 * it), benchmark_sample_from=int(500 * it)), 152))
    # config 3 (8 GPUs) is bench.py --gpus 8; here its per-GPU shape for both decompositions, 2 subdomains
    # of the reference's x-split layout on this one GPU (halo path exercised, no xGMI)
    for cid, pattern in (('3', 'AA'), ('3b', 'AB')):
        res.append(run('%s: D3Q19 BGK 1024x512x512 / 8: one 128x512x512 x-slab pair on one GPU (%s)' % (cid, pattern), BoxSim,
                       EqualSubdomainsGeometry3D,
                       dict(lat_nx=256, lat_ny=512, lat_nz=512, subdomains=2, conn_axis='x', periodic_x=True,
                            periodic_y=True, periodic_z=True, visc=1.0 / 6.0, access_pattern=pattern, grid='D3Q19',
                            max_iters=int(600 * it), benchmark_sample_from=int(200 * it)), 152))
    # ... and the whole of config 3 in ONE process on this GPU: --subdomains=8 the way the reference runs it on a single
    # device (controller.LocalGroup: eight runners stepped in lock-step from one group plan, x-face buffers between them)
    res.append(run('3g8: D3Q19 BGK 1024x512x512 in 8 x-slabs, all in one process on one GPU (AA)', BoxSim,
                   EqualSubdomainsGeometry3D,
                   dict(lat_nx=1024, lat_ny=512, lat_nz=512, subdomains=8, conn_axis='x', periodic_x=True,
                        periodic_y=True, periodic_z=True, visc=1.0 / 6.0, access_pattern='AA', grid='D3Q19',
                        max_iters=int(300 * it), benchmark_sample_from=int(100 * it)), 152))
    # config 3 AS STATED: all eight subdomains of 1024 x 512 x 512, one process each.  On a 1-GPU box the eight ranks share
    # the device (gloo group for the rendezvous; the halos go through the peer transport: the neighbours' receive buffers
    # mapped into every process, sailfish_amd/peer.py -- a RATE since round 6, not only a functional run); with eight GPUs
    # visible the same call runs one rank per GPU.  bench.py does the work (and validates the seams + whole planes of
    # the undivided box).
    for cid, axis in (('3x8', 'x'), ('3z8', 'z')):
        if only and cid not in only:
            continue
        import subprocess
        import torch
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        if torch.cuda.device_count() < 8:
            env.update(SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0')
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--scaling', 'strong', '--domain', '1024x512x512',
               '--axis', axis, '--steps', '20', '--warmup', '5', '--prewarm_steps', '20', '--repeats', '1', '--no_cpu_baseline',
               '--no_gpu_state', '--min_seconds', '0.1', '--halo_timing_steps', '4']
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env).stdout.decode(errors='replace')
        lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
        if lines:
            d = json.loads(lines[-1])
            c = d['config']
            r = {'config': '%s: D3Q19 BGK 1024x512x512 in 8 %s-slabs, one process each (%s)' % (cid, axis, c['dist_backend']),
                 'MLUPS_eff': d['value'], 'ranks': d['n_gpus'], 'rccl_ranks': c['rccl_ranks'], 'validated': c.get('validated'),
                 'halo_transport': (c.get('halo_transport') or '').split(':')[0], 'ms_per_step': d['ms_per_step'],
                 'frac_of_8TBps': round(d['value'] * 152 / 8e6, 4),
                 'undivided_box_bit_identical': all(v.get('undivided_box', {}).get('populations_bit_identical') for v in c.get('validation', {}).values()),
                 'halo_overlap_frac': c.get('halo_overlap_frac'), 'access_pattern': c['access_pattern']}
            print(json.dumps(r), flush=True)
            res.append(r)
    # config 4: binary Shan-Chen 256^3 (3 passes per step: 516 B per node update)
    res.append(run('4: binary Shan-Chen D3Q19 256^3', SeparationSim, LBGeometry3D,
                   dict(lat_nx=256, lat_ny=256, lat_nz=256, access_pattern='AA', max_iters=int(1500 * it),
                        benchmark_sample_from=int(500 * it)), 516))
    # not BASELINE configurations (only with --only): an open duct, inlet + do-nothing outlet on the x faces / the z faces
    for cid, sim_cls, pattern, per in (('6xa', DuctSimX, 'AA', dict(periodic_z=True)), ('6xb', DuctSimX, 'AB', dict(periodic_z=True)),
                                       ('6za', DuctSimZ, 'AA', dict(periodic_x=True)), ('6zb', DuctSimZ, 'AB', dict(periodic_x=True))):
        if cid in only:
            res.append(run('%s: D3Q19 BGK open duct 512x256x256, inlet / do-nothing outlet on the %s faces (%s)' % (cid, cid[1], pattern),
                           sim_cls, LBGeometry3D,
                           dict(lat_nx=512, lat_ny=256, lat_nz=256, visc=0.05, access_pattern=pattern, max_iters=int(1500 * it),
                                benchmark_sample_from=int(500 * it), **per), 152))
    # ---- round 6: the rows of SURVEY 8(f) that had no number.  Not BASELINE configurations (only with --only).
    # (f)4 --node_addressing=indirect (reference subdomain_runner.py:829-878, kernel_common.mako:140-167): a packed bed of
    # random spheres, ~30 % fluid, and the wavy pipe of examples/external_geometry.py scaled up; the same geometries with
    # direct addressing beside them.  Bytes per ACTIVE node update: the 2 Q populations + the node map and the dense
    # node -> slot table (4 B each per DENSE node, read once from HBM: their neighbours' entries come from the caches),
    # spread over the active nodes.
    from examples.external_geometry import GeometrySim

    class PorousSim(GeometrySim):
        @classmethod
        def modify_config(cls, config):
            import numpy as np
            wall = porous_wall_map(config.lat_nx, config.lat_ny, config.lat_nz)
            config._wall_map = np.pad(wall, 1, 'wrap')
            config.periodic_x = config.periodic_y = config.periodic_z = True

    def indirect_bytes(out):
        return 152 + 8.0 / max(out['fluid_fraction'], 1e-6)

    def direct_bytes(out):        # the dense kernels move every node's populations only where the node is active; + the map
        return 152 + 4.0 / max(out['fluid_fraction'], 1e-6)
    for cid, sim_cls, addr, dims, what in (
            ('7a', PorousSim, 'indirect', (512, 256, 256), 'packed bed of random spheres'),
            ('7b', PorousSim, 'direct', (512, 256, 256), 'packed bed of random spheres'),
            ('7c', GeometrySim, 'indirect', (512, 161, 161), 'wavy pipe of examples/external_geometry.py'),
            ('7d', GeometrySim, 'direct', (512, 161, 161), 'wavy pipe of examples/external_geometry.py')):
        if cid in only:
            res.append(run('%s: D3Q19 BGK %s %dx%dx%d, --node_addressing=%s (AA)' % ((cid, what) + dims + (addr,)), sim_cls, LBGeometry3D,
                           dict(lat_nx=dims[0], lat_ny=dims[1], lat_nz=dims[2], visc=0.05, access_pattern='AA', node_addressing=addr,
                                max_iters=int(1500 * it), benchmark_sample_from=int(500 * it)),
                           indirect_bytes if addr == 'indirect' else direct_bytes))
    # (f)3 single-component Shan-Chen (reference lb_single.py:242-347: PrepareMacroFields + CollideAndPropagate): 3 Q + 2
    # values per update -- the density pass reads Q and writes 1, the sweep reads Q + 1 and writes Q
    from examples.sc_phase_separation import PhaseSeparationSim, VapourSubdomain
    from sailfish.subdomain import Subdomain3D as _S3

    class Vapour3D(_S3):
        def boundary_conditions(self, hx, hy, hz):
            pass

        def initial_conditions(self, sim, hx, hy, hz):
            import numpy as np
            rng = np.random.RandomState(self.config.seed)
            sim.rho[:] = 0.693 + rng.rand(*sim.rho.shape) / 100

    class PhaseSeparation3D(PhaseSeparationSim):
        subdomain = Vapour3D

    if '8a' in only:
        res.append(run('8a: single-component Shan-Chen D3Q19 256^3 (AA)', PhaseSeparation3D, LBGeometry3D,
                       dict(lat_nx=256, lat_ny=256, lat_nz=256, grid='D3Q19', periodic_x=True, periodic_y=True, periodic_z=True,
                            access_pattern='AA', seed=11, max_iters=int(1500 * it), benchmark_sample_from=int(500 * it)), 3 * 76 + 8))
    if '8b' in only:
        res.append(run('8b: single-component Shan-Chen D2Q9 1024^2 (AA)', PhaseSeparationSim, LBGeometry2D,
                       dict(lat_nx=1024, lat_ny=1024, periodic_x=True, periodic_y=True, access_pattern='AA', seed=11,
                            max_iters=int(20000 * it), benchmark_sample_from=int(5000 * it)), 3 * 36 + 8))
    # the reference's obstacle examples at a size that fills the device: bounce-back cylinder / sphere in a channel
    from examples.cylinder import CylinderSim
    from examples.sphere_3d import SphereSim
    if '9a' in only:
        res.append(run('9a: examples/cylinder.py D2Q9 BGK 2048x1024 (AA)', CylinderSim, LBGeometry2D,
                       dict(lat_nx=2048, lat_ny=1024, access_pattern='AA', max_iters=int(8000 * it),
                            benchmark_sample_from=int(2000 * it)), 72))
    if '9b' in only:
        res.append(run('9b: examples/sphere_3d.py D3Q19 BGK 512x256x256 (AA)', SphereSim, LBGeometry3D,
                       dict(lat_nx=512, lat_ny=256, lat_nz=256, access_pattern='AA', max_iters=int(1500 * it),
                            benchmark_sample_from=int(500 * it)), 152))
    # ---- several x-slabs of ONE process on this GPU (controller.LocalGroup; round 6): the Shan-Chen models over the x-face
    # planes, the force-driven pipe with its cuts on 128-byte lines
    if '4x4' in only:
        res.append(run('4x4: binary Shan-Chen D3Q19 256^3 in 4 x-slabs, one process (AB)', SeparationSim, EqualSubdomainsGeometry3D,
                       dict(lat_nx=256, lat_ny=256, lat_nz=256, subdomains=4, conn_axis='x', access_pattern='AB',
                            max_iters=int(900 * it), benchmark_sample_from=int(300 * it)), 516))
    if '8x4' in only:
        res.append(run('8x4: single-component Shan-Chen D3Q19 256^3 in 4 x-slabs, one process (AA)', PhaseSeparation3D,
                       EqualSubdomainsGeometry3D,
                       dict(lat_nx=256, lat_ny=256, lat_nz=256, grid='D3Q19', periodic_x=True, periodic_y=True, periodic_z=True,
                            subdomains=4, conn_axis='x', access_pattern='AA', seed=11, max_iters=int(1500 * it),
                            benchmark_sample_from=int(500 * it)), 3 * 76 + 8))
    if '5x3' in only:
        res.append(run('5x3: D3Q19 BGK force-driven pipe 512x256x256 in 3 x-slabs, one process (AA)', Pipe3D,
                       EqualSubdomainsGeometry3D,
                       dict(lat_nx=512, lat_ny=256, lat_nz=256, visc=0.05, subdomains=3, conn_axis='x', access_pattern='AA',
                            max_iters=int(1500 * it), benchmark_sample_from=int(500 * it)), 152))
    if args.out:
        with open(args.out, 'w') as fh:
            for r in filter(None, res):
                fh.write(json.dumps(r) + '\n')


if __name__ == '__main__':
    main()
