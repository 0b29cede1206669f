#!/usr/bin/env python3
"""HBM write / read bandwidth of the simplest possible kernels as a function of WHERE the bytes go (GPU box):
  scan:   one contiguous span of --span_gib written / read / updated in place at offsets k GiB of a big block
  spread: the same number of bytes split into K streams S GiB apart (K x S table)
Tells whether the sweep's fast / slow placement modes are a property of plain writes to a memory region.
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from sailfish_amd.backend_hip import HIPBackend
from tools.box_probe import build_probe_lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--block_gib', type=int, default=200)
    ap.add_argument('--span_gib', type=float, default=10.0)
    ap.add_argument('--scan_step', type=float, default=2.0)
    ap.add_argument('--reps', type=int, default=4)
    args = ap.parse_args()

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    lib = ctypes.CDLL(build_probe_lib())
    lib.probe_kstream.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t,
                                  ctypes.c_void_p, ctypes.c_void_p]
    G = 1 << 30
    block = b.alloc_buf(size=args.block_gib * G + (4 << 20))
    base = (block + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    sink = b.alloc_buf(size=4096)
    stream = b.make_stream()
    native = ctypes.c_void_p(stream.native)

    def run(kind, off, bytes_per_stream, K, S):
        def launch():
            rc = lib.probe_kstream(kind, ctypes.c_void_p(base + off), bytes_per_stream, K, S, ctypes.c_void_p(sink), native)
            assert rc == 0, rc
        launch()
        e0 = b.make_event(stream, timing=True)
        for _ in range(args.reps):
            launch()
        e1 = b.make_event(stream, timing=True)
        e1.synchronize()
        ms = e1.time_since(e0) / args.reps
        return bytes_per_stream * K * (2 if kind == 2 else 1) / ms / 1e6

    span = int(args.span_gib * G)
    print('block %d GiB at 0x%x' % (args.block_gib, base))
    print('== scan: one span of %.1f GiB at offset (GiB): write / read / in-place update, GB/s' % args.span_gib)
    k = 0.0
    while k + args.span_gib <= args.block_gib:
        off = int(k * G)
        print('  off %6.1f  write %6.0f  read %6.0f  rmw %6.0f' % (k, run(0, off, span, 1, 0), run(1, off, span, 1, 0), run(2, off, span, 1, 0)), flush=True)
        k += args.scan_step
    print('== spread: %.1f GiB in K streams S GiB apart (first stream at offset 0): write GB/s | rmw GB/s' % args.span_gib)
    for K in (1, 2, 4, 8, 16):
        row = []
        for S in (1, 2, 4, 8, 12, 16, 24, 32):
            if (K - 1) * S * G + span // K > args.block_gib * G:
                row.append('    -      ')
                continue
            row.append('%5.0f|%5.0f' % (run(0, 0, span // K // 4096 * 4096, K, S * G), run(2, 0, span // K // 4096 * 4096, K, S * G)))
        print('  K=%2d  ' % K + '  '.join(row), flush=True)
    print('        S= 1           2           4           8          12          16          24          32')


if __name__ == '__main__':
    main()
