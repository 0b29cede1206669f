#!/usr/bin/env python3
"""D2Q9 sweep rate on large 2-D boxes (GPU box only): periodic box through BoxSim, HIP events.

    python tools/probe/d2q9_probe.py --sizes 1024,4096,8192 [--general]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from sailfish_amd import sym  # noqa: E402
from sailfish_amd.backend_hip import HIPBackend  # noqa: E402
from sailfish_amd.box import BoxSim, make_box_desc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='1024,4096,8192')
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--models', default='bgk,mrt')
    args = ap.parse_args()

    class Opt(object):
        pass
    b = HIPBackend(Opt(), 0)
    for n in (int(x) for x in args.sizes.split(',')):
        for model in args.models.split(','):
            for pattern in ('AA', 'AB'):
                desc = make_box_desc(sym.D2Q9, (n, n), model=model, precision='single', access_pattern=pattern,
                                     visc=1.0 / 6.0, periodic_fused=[1, 1, 0])
                s = BoxSim(b, desc, periodic=(True, True, False))
                y, x = np.mgrid[0:n, 0:n].astype(np.float32)
                s.set_fields(np.ones((n, n), np.float32), [0.05 * np.sin(2 * np.pi * y / n), 0.05 * np.sin(2 * np.pi * x / n)])
                s.initial_conditions()
                for _ in range(10):
                    s.step()
                s.sync()
                e0 = b.make_event(s.stream, timing=True)
                for _ in range(args.reps):
                    s.step()
                e1 = b.make_event(s.stream, timing=True)
                e1.synchronize()
                ms = e1.time_since(e0) / args.reps
                print('D2Q9 %s %s %5d^2  %.4f ms/step  %8.1f MLUPS  %6.0f GB/s (72 B/update)'
                      % (model, pattern, n, ms, n * n / ms * 1e-3, n * n * 72 / ms * 1e-6), flush=True)
                s.release()


if __name__ == '__main__':
    main()
