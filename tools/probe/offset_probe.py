#!/usr/bin/env python3
"""Where inside ONE big allocation is the sweep fast, where slow?  The distribution array (10.9 GB at 512^3) is placed at
offsets k * step inside a single hipMalloc'ed block and the in-place (AA) kernels are timed at every offset.  A periodic
pattern over the offset exposes which address bits decide the mode.

    python tools/probe/offset_probe.py --dims 512x512x512 --steps_mib 2,128 --count 48
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
    ap.add_argument('--steps_mib', default='2,128')
    ap.add_argument('--count', type=int, default=48)
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--modes', default='even')
    args = ap.parse_args()

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    size = tuple(int(x) for x in args.dims.split('x'))
    grid = sym.D3Q19
    desc = make_box_desc(grid, size, precision='single', access_pattern='AA', visc=1.0 / 6.0, periodic_fused=[1, 1, 1])
    nbytes = 19 * hipabi.dist_stride(desc) * 4
    steps = [int(x) for x in args.steps_mib.split(',')]
    span = max(steps) * args.count << 20
    block = b.alloc_buf(size=nbytes + span + (4 << 20))        # zero-filled by alloc_buf
    base = (block + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    off0 = b.dist_align_offset(4)
    shape = (desc.arr_nz, desc.arr_ny, desc.arr_nx)
    rho = (1.0 + 1e-3 * np.random.RandomState(1).rand(*shape)).astype(np.float32)
    v = np.zeros(shape, dtype=np.float32)
    g_rho = b.alloc_buf(like=rho)
    g_v = [b.alloc_buf(like=v) for _ in range(3)]
    stream = b.make_stream()
    mod = b.build(desc)
    sig = 'PPPPPPPi'
    bytes_step = size[0] * size[1] * size[2] * 152
    print('block 0x%x (%.1f GB), 2 MiB-aligned base 0x%x, array %.2f GB' % (block, (nbytes + span) / 1e9, base, nbytes / 1e9))
    # a benign state everywhere the arrays may land: rest-state populations are not needed for timing, zeros collide to zeros
    for step in steps:
        line = []
        for k in range(args.count):
            d = base + (k * step << 20) + off0
            kern = b.get_kernel(mod, 'CollideAndPropagate', (64,), [0, d, d, g_rho] + g_v + [0], sig, needs_iteration=True)
            res = []
            for mode in args.modes.split(','):
                it = 0 if mode == 'even' else 1
                b._lib.slf_kernel_set_iteration(kern.handle, it)
                for _ in range(3):
                    b.run_kernel(kern, None, stream)
                e0 = b.make_event(stream, timing=True)
                for _ in range(args.reps):
                    b.run_kernel(kern, None, stream)
                e1 = b.make_event(stream, timing=True)
                e1.synchronize()
                res.append(e1.time_since(e0) / args.reps)
            line.append(res)
        print('step %d MiB:' % step)
        for k, res in enumerate(line):
            print('  off %6d MiB  %s  %s' % (k * step, ' '.join('%.3f' % t for t in res), 'SLOW' if res[0] > 3.6 * bytes_step / 20401094656.0 else ''))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
