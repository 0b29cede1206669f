#!/usr/bin/env python3
"""Does the rate of a placed array depend on what the process allocated and freed before?  (GPU box only.)
Alternates a small placed box (256^3, or two 128x512x512 slabs) with a large one (512^3 AB) that is released again."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from sailfish_amd import sym  # noqa: E402
from sailfish_amd.backend_hip import HIPBackend  # noqa: E402
from sailfish_amd.slab import SlabSim  # noqa: E402


class Opt(object):
    pass


def rate(b, size, pattern, steps=200, same_backend=True):
    s = SlabSim(b, sym.D3Q19, size, access_pattern=pattern, visc=1.0 / 6.0)
    s.init_synthetic(seed=1)
    for _ in range(50):
        s.step()
    s.sync()
    e0 = b.make_event(s.calc_stream, timing=True)
    for _ in range(steps):
        s.step()
    e1 = b.make_event(s.calc_stream, timing=True)
    e1.synchronize()
    ms = e1.time_since(e0) / steps
    n = size[0] * size[1] * size[2]
    info = s.placement_info
    s.release()
    return n * 152 / ms * 1e-6, info


def main():
    shared = HIPBackend(Opt(), 0)
    for rnd in range(4):
        for size in ((256, 256, 256), (128, 512, 512)):
            b = shared if os.environ.get('SEQ_SHARED', '1') == '1' else HIPBackend(Opt(), 0)
            gbs, info = rate(b, size, 'AA')
            print('round %d  %-12s AA %6.0f GB/s  %s' % (rnd, 'x'.join(map(str, size)), gbs, info), flush=True)
        b = shared if os.environ.get('SEQ_SHARED', '1') == '1' else HIPBackend(Opt(), 0)
        gbs, info = rate(b, (512, 512, 512), 'AB', steps=60)
        print('round %d  512^3        AB %6.0f GB/s  %s' % (rnd, gbs, info), flush=True)


if __name__ == '__main__':
    main()
