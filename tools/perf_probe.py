#!/usr/bin/env python3
"""Kernel-variant probe (GPU box only): times the even / odd AA sweep and the AB sweep of the
D3Q19 periodic box for several SLF_VARIANT / SLF_BLOCK_X settings with HIP events.

    python tools/perf_probe.py --size 512 --variants 0,1,2,4,5 --blocks 576,256
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from sailfish_amd import sym
from sailfish_amd.backend_hip import HIPBackend
from sailfish_amd.box import make_box_desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--dims', default='', help='NXxNYxNZ instead of --size (cube)')
    ap.add_argument('--variants', default='0,1,4,5')
    ap.add_argument('--blocks', default='576,256')
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--model', default='bgk')
    ap.add_argument('--modes', default='even,odd,aa,ab')
    ap.add_argument('--norelax', action='store_true')
    ap.add_argument('--noalign', action='store_true')
    ap.add_argument('--pads', default='0', help='comma list of dist_stride paddings (elements)')
    ap.add_argument('--general', default='', choices=['', 'fluid', 'walls', 'pipe'],
                    help='run the node-map (general) kernels: all-fluid periodic box / closed box of full-BB walls / '
                         'circular pipe along z (unused nodes outside)')
    ap.add_argument('--accel', type=float, default=0.0, help='body-force acceleration along x (FORCE instantiations)')
    ap.add_argument('--plain_types', action='store_true',
                    help='--general: a type table without boundary-condition kinds (Geometry::bc_level 0)')
    ap.add_argument('--trace', type=int, default=0, help='also print the time of every batch of N launches')
    args = ap.parse_args()

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    n = args.size
    grid = sym.D3Q19
    size = (n, n, n)
    if args.dims:
        size = tuple(int(x) for x in args.dims.split('x'))
        assert not args.general, '--general needs a cube'
    desc0 = make_box_desc(grid, size, model=args.model, precision='single', access_pattern='AA', visc=1.0 / 6.0,
                          periodic_fused=[1, 1, 1], dist_pad=0)
    nodes = desc0.arr_nx * desc0.arr_ny * desc0.arr_nz
    pads = [int(x) for x in args.pads.split(',')]
    maxstride = nodes + max(pads)
    off = b.dist_align_offset(4) if not args.noalign else 0
    from sailfish_amd import placement
    if placement.enabled() and 19 * maxstride * 4 >= placement.MIN_BYTES:
        bufs = b.alloc_placed([19 * maxstride * 4] * (2 if 'ab' in args.modes else 1), off)
        dist_a = bufs[0].addr
        dist_b = bufs[1].addr if len(bufs) > 1 else 0
    else:
        dist_a = b.alloc_buf(size=19 * maxstride * 4, align_offset=off)
        dist_b = b.alloc_buf(size=19 * maxstride * 4, align_offset=off) if 'ab' in args.modes else 0
    print('dist_a 0x%x dist_b 0x%x stride_bytes %d' % (dist_a, dist_b, maxstride * 4))
    shape = (desc0.arr_nz, desc0.arr_ny, desc0.arr_nx)
    rho = np.ones(shape, dtype=np.float32)
    rho += (1e-3 * np.random.RandomState(1).rand(*shape)).astype(np.float32)
    v = np.zeros(shape, dtype=np.float32)
    g_rho = b.alloc_buf(like=rho)
    g_v = [b.alloc_buf(like=v) for _ in range(3)]
    stream = b.make_stream()
    mod0 = b.build(desc0)
    gkw, g_map, fused = {}, 0, [1, 1, 1]
    if args.general:
        from tests import _geometry as geo
        m = geo.empty_map(desc0)
        if args.general == 'walls':
            fused = [0, 0, 0]
            w = geo.encode(geo.T_FULLBB)
            m[1, 1:n + 1, 1:n + 1] = w
            m[n, 1:n + 1, 1:n + 1] = w
            m[1:n + 1, 1, 1:n + 1] = w
            m[1:n + 1, n, 1:n + 1] = w
            m[1:n + 1, 1:n + 1, 1] = w
            m[1:n + 1, 1:n + 1, n] = w
        elif args.general == 'pipe':
            fused = [0, 0, 1]
            yy, xx = np.mgrid[0:desc0.arr_ny, 0:desc0.arr_nx]
            r2 = (yy - (n + 1) / 2.0) ** 2 + (xx - (n + 1) / 2.0) ** 2
            inside = r2 < (n / 2.0 - 1) ** 2
            ring = (~inside) & (r2 < (n / 2.0 + 0.5) ** 2)
            real = np.zeros_like(inside)
            real[1:n + 1, 1:n + 1] = True
            m[1:n + 1, (~inside & ~ring) & real] = geo.encode(geo.T_UNUSED)
            m[1:n + 1, ring & real] = geo.encode(geo.T_FULLBB)
        g_map = b.alloc_buf(like=np.ascontiguousarray(m))
        kinds = list(geo.TYPE_KIND)
        if args.plain_types:
            from sailfish_amd import hipabi as h
            plain = (h.SLF_NK_FLUID, h.SLF_NK_GHOST, h.SLF_NK_FULL_BB, h.SLF_NK_UNUSED)
            kinds = [k if k in plain else h.SLF_NK_UNUSED for k in kinds]
        gkw = dict(fluid_only=False, type_kind=kinds, nt_bits=geo.NT_BITS)
        print('general map %s: %.1f%% of the real nodes excluded' % (
            args.general, 100.0 * np.mean((m[1:n + 1, 1:n + 1, 1:n + 1] & 7) == geo.T_UNUSED)))
    for d in ([dist_a, dist_b] if dist_b else [dist_a]):
        k = b.get_kernel(mod0, 'SetInitialConditions', (64,), [d] + g_v + [g_rho, 0], 'PPPPPP')
        b.run_kernel(k, None, stream)
    stream.synchronize()
    bytes_step = size[0] * size[1] * size[2] * 152
    print('size %s, arr_nx %d, %d reps' % ('x'.join(map(str, size)), desc0.arr_nx, args.reps))
    for variant in [int(x) for x in args.variants.split(',')]:
      for pad in pads:
        for bx in [int(x) for x in args.blocks.split(',')]:
            os.environ['SLF_VARIANT'] = str(variant)
            os.environ['SLF_BLOCK_X'] = str(bx)
            res = []
            for mode in args.modes.split(','):
                ap_ = 'AB' if mode == 'ab' else 'AA'
                desc = make_box_desc(grid, size, model=args.model, precision='single', access_pattern=ap_,
                                     visc=1.0 / 6.0, periodic_fused=fused,
                                     relaxation_enabled=not args.norelax, dist_pad=pad,
                                     accel=[args.accel, 0.0, 0.0] if args.accel else None, **gkw)
                mod = b.build(desc)
                sig = 'PPPPPPPi'
                for dd in ([dist_a, dist_b] if dist_b else [dist_a]):
                    kk = b.get_kernel(mod, 'SetInitialConditions', (64,), [dd] + g_v + [g_rho, 0], 'PPPPPP')
                    b.run_kernel(kk, None, stream)
                if mode == 'ab':
                    ks = [b.get_kernel(mod, 'CollideAndPropagate', (64,), [g_map, dist_a, dist_b, g_rho] + g_v + [0], sig),
                          b.get_kernel(mod, 'CollideAndPropagate', (64,), [g_map, dist_b, dist_a, g_rho] + g_v + [0], sig)]
                else:
                    ks = [b.get_kernel(mod, 'CollideAndPropagate', (64,), [g_map, dist_a, dist_a, g_rho] + g_v + [0], sig,
                                       needs_iteration=True)]

                def launch(i):
                    if mode == 'ab':
                        b.run_kernel(ks[i & 1], None, stream)
                    else:
                        it = {'even': 0, 'odd': 1, 'aa': i}[mode]
                        b._lib.slf_kernel_set_iteration(ks[0].handle, it)
                        b.run_kernel(ks[0], None, stream)
                for i in range(4):
                    launch(i)
                evs = [b.make_event(stream, timing=True)]
                for i in range(args.reps):
                    launch(i)
                    if args.trace and (i + 1) % args.trace == 0:
                        evs.append(b.make_event(stream, timing=True))
                e1 = b.make_event(stream, timing=True)
                e1.synchronize()
                ms = e1.time_since(evs[0]) / args.reps
                if args.trace:
                    print('   trace %s:' % mode, ' '.join('%.2f' % (evs[j + 1].time_since(evs[j]) / args.trace)
                                                           for j in range(len(evs) - 1)))
                res.append('%s %.3f ms %.0f GB/s' % (mode, ms, bytes_step / ms / 1e6))
            print('variant %2d pad %6d block %4d | %s' % (variant, pad, bx, ' | '.join(res)), flush=True)


if __name__ == '__main__':
    main()
