#!/usr/bin/env python3
"""Is the fast / slow mode of the sweep a property of the PROCESS or of the ALLOCATION?  One process, K separately
allocated distribution arrays (each 19 x 544 x 514 x 514 x 4 B = 10.9 GB at 512^3): the same even / odd AA kernels are
timed on every array in turn (several rounds), then the AB kernel on pairs.  Optional gaps of G bytes are allocated
between the arrays so that they land at different offsets.

    python tools/probe/alloc_probe.py --dims 512x512x512 --arrays 6 --rounds 3
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from sailfish_amd import sym
from sailfish_amd.backend_hip import HIPBackend
from sailfish_amd.box import make_box_desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dims', default='512x512x512')
    ap.add_argument('--arrays', type=int, default=6)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--gaps', default='0', help='comma list of gap sizes in MiB allocated before array i (cycled)')
    ap.add_argument('--align_mib', type=int, default=0, help='round every array start up to this many MiB')
    args = ap.parse_args()

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    size = tuple(int(x) for x in args.dims.split('x'))
    grid = sym.D3Q19
    desc = make_box_desc(grid, size, precision='single', access_pattern='AA', visc=1.0 / 6.0, periodic_fused=[1, 1, 1])
    desc_ab = make_box_desc(grid, size, precision='single', access_pattern='AB', visc=1.0 / 6.0, periodic_fused=[1, 1, 1])
    from sailfish_amd import hipabi
    nodes = desc.arr_nx * desc.arr_ny * desc.arr_nz
    nbytes = 19 * hipabi.dist_stride(desc) * 4       # the stride includes SLF_DIST_PAD
    off = b.dist_align_offset(4)
    gaps = [int(x) for x in args.gaps.split(',')]
    arrays, keep = [], []
    for i in range(args.arrays):
        g = gaps[i % len(gaps)]
        if g:
            keep.append(b.alloc_buf(size=g << 20))
        if args.align_mib:
            a = args.align_mib << 20
            raw = b.alloc_buf(size=nbytes + a + 256)
            base = (raw + a - 1) // a * a
            arrays.append(base + off)
        else:
            arrays.append(b.alloc_buf(size=nbytes, align_offset=off))
    shape = (desc.arr_nz, desc.arr_ny, desc.arr_nx)
    rho = (1.0 + 1e-3 * np.random.RandomState(1).rand(*shape)).astype(np.float32)
    v = np.zeros(shape, dtype=np.float32)
    g_rho = b.alloc_buf(like=rho)
    g_v = [b.alloc_buf(like=v) for _ in range(3)]
    stream = b.make_stream()
    mod, mod_ab = b.build(desc), b.build(desc_ab)
    for d in arrays:
        k = b.get_kernel(mod, 'SetInitialConditions', (64,), [d] + g_v + [g_rho, 0], 'PPPPPP')
        b.run_kernel(k, None, stream)
    stream.synchronize()
    sig = 'PPPPPPPi'
    bytes_step = size[0] * size[1] * size[2] * 152

    def time_kernel(k, its):
        for it in its[:4]:
            if it is not None:
                b._lib.slf_kernel_set_iteration(k.handle, it)
            b.run_kernel(k, None, stream)
        e0 = b.make_event(stream, timing=True)
        for i in range(args.reps):
            it = its[i % len(its)]
            if it is not None:
                b._lib.slf_kernel_set_iteration(k.handle, it)
            b.run_kernel(k, None, stream)
        e1 = b.make_event(stream, timing=True)
        e1.synchronize()
        return e1.time_since(e0) / args.reps

    print('dims %s, %d arrays of %.2f GB' % (args.dims, len(arrays), nbytes / 1e9))
    kaa = [b.get_kernel(mod, 'CollideAndPropagate', (64,), [0, d, d, g_rho] + g_v + [0], sig, needs_iteration=True) for d in arrays]
    for r in range(args.rounds):
        for i, d in enumerate(arrays):
            te = time_kernel(kaa[i], [0])
            to = time_kernel(kaa[i], [1])
            print('round %d array %d va 0x%012x (MiB off in 1 GiB: %7.2f) | even %.3f ms %.0f GB/s | odd %.3f ms %.0f GB/s'
                  % (r, i, d, (d % (1 << 30)) / 2.0 ** 20, te, bytes_step / te / 1e6, to, bytes_step / to / 1e6), flush=True)
    for i in range(len(arrays)):
        j = (i + 1) % len(arrays)
        k = b.get_kernel(mod_ab, 'CollideAndPropagate', (64,), [0, arrays[i], arrays[j], g_rho] + g_v + [0], sig)
        t = time_kernel(k, [None])
        print('AB %d -> %d | %.3f ms %.0f GB/s' % (i, j, t, bytes_step / t / 1e6), flush=True)


if __name__ == '__main__':
    main()
