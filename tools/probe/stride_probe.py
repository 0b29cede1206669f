#!/usr/bin/env python3
"""On the SAME physical region, which stride between the 19 direction arrays makes the sweep fast?  One big block; for
every offset (GiB) in --offsets the distribution array is placed there and the in-place even/odd kernels are timed for
every dist_stride padding in --pads (elements added to arr_nx*arr_ny*arr_nz).

    python tools/probe/stride_probe.py --offsets 0,8,27 --pads 0,32,1024,65536
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from sailfish_amd import hipabi, sym
from sailfish_amd.backend_hip import HIPBackend
from sailfish_amd.box import make_box_desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dims', default='512x512x512')
    ap.add_argument('--offsets', default='0,8,16,24,28,32')
    ap.add_argument('--pads', default='0,32,1024,32768,65536,131072,262144,524288,1048576,2097152')
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--mode', default='even')
    ap.add_argument('--ab_interleave', action='store_true', help='mode ab: copy B sits in the gaps of copy A (half a stride later)')
    args = ap.parse_args()

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    size = tuple(int(x) for x in args.dims.split('x'))
    grid = sym.D3Q19
    pads = [int(x) for x in args.pads.split(',')]
    offsets = [float(x) for x in args.offsets.split(',')]
    desc0 = make_box_desc(grid, size, precision='single', access_pattern='AA', visc=1.0 / 6.0, periodic_fused=[1, 1, 1], dist_pad=0)
    nodes = desc0.arr_nx * desc0.arr_ny * desc0.arr_nz
    nbytes_max = 19 * (nodes + max(pads)) * 4
    span = int(max(offsets) * (1 << 30))
    two = 2 if (args.mode == 'ab' and not args.ab_interleave) else 1
    block = b.alloc_buf(size=two * (nbytes_max + (2 << 20)) + span + (4 << 20))
    base = (block + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    off0 = b.dist_align_offset(4)
    shape = (desc0.arr_nz, desc0.arr_ny, desc0.arr_nx)
    rho = (1.0 + 1e-3 * np.random.RandomState(1).rand(*shape)).astype(np.float32)
    v = np.zeros(shape, dtype=np.float32)
    g_rho = b.alloc_buf(like=rho)
    g_v = [b.alloc_buf(like=v) for _ in range(3)]
    stream = b.make_stream()
    sig = 'PPPPPPPi'
    print('block %.1f GB at 0x%x; array %.2f GB; offsets (GiB) %s' % ((nbytes_max + span) / 1e9, base, 19 * nodes * 4 / 1e9, offsets))
    print('%10s ' % 'pad' + ' '.join('%8.1f' % o for o in offsets))
    for pad in pads:
        desc = make_box_desc(grid, size, precision='single', access_pattern='AA', visc=1.0 / 6.0, periodic_fused=[1, 1, 1],
                             dist_pad=pad) if pad else desc0
        mod = b.build(desc)
        desc_ab = make_box_desc(grid, size, precision='single', access_pattern='AB', visc=1.0 / 6.0,
                                periodic_fused=[1, 1, 1], dist_pad=pad)
        mod_ab = b.build(desc_ab)
        row = []
        for o in offsets:
            d = base + (int(o * 1024) << 20) + off0
            ki = b.get_kernel(mod, 'SetInitialConditions', (64,), [d] + g_v + [g_rho, 0], 'PPPPPP')
            b.run_kernel(ki, None, stream)
            if args.mode == 'ab':
                d2 = d + (nbytes_max + (2 << 20)) // (2 << 20) * (2 << 20)
                if args.ab_interleave:
                    d2 = d + ((nodes + pad) // 2 * 4) // (2 << 20) * (2 << 20)
                    assert (nodes + pad) // 2 * 4 >= nodes * 4 + (4 << 20), 'gaps too small for copy B'
                ki2 = b.get_kernel(mod, 'SetInitialConditions', (64,), [d2] + g_v + [g_rho, 0], 'PPPPPP')
                b.run_kernel(ki2, None, stream)
                ks = [b.get_kernel(mod_ab, 'CollideAndPropagate', (64,), [0, d, d2, g_rho] + g_v + [0], sig),
                      b.get_kernel(mod_ab, 'CollideAndPropagate', (64,), [0, d2, d, g_rho] + g_v + [0], sig)]
            else:
                kern = b.get_kernel(mod, 'CollideAndPropagate', (64,), [0, d, d, g_rho] + g_v + [0], sig, needs_iteration=True)
                b._lib.slf_kernel_set_iteration(kern.handle, 0 if args.mode == 'even' else 1)
                ks = [kern, kern]
            for i in range(2):
                b.run_kernel(ks[i & 1], None, stream)
            e0 = b.make_event(stream, timing=True)
            for i in range(args.reps):
                b.run_kernel(ks[i & 1], None, stream)
            e1 = b.make_event(stream, timing=True)
            e1.synchronize()
            row.append(e1.time_since(e0) / args.reps)
        print('%10d ' % pad + ' '.join('%8.3f' % t for t in row), flush=True)


if __name__ == '__main__':
    main()
