"""What the x-face buffers cost the sweep of a 128 x 512 x 512 slab (BASELINE config 4's subdomain): sweep-only launches
(no exchange) of SlabSim with connected x faces, with the edge lanes' stores into the send planes and / or their loads from
the receive planes switched off (probe bits 2048 / 4096 of SLF_VARIANT; results are then wrong, only the time counts),
next to the same slab as a periodic box.  HIP events, K steps per sample, median of N samples.

    python tools/probe/xface_cost_probe.py [AA|AB] [steps] [samples]

The probe bits are NOT in the shipped kernels: they were three one-line guards (`if (g.variant & 2048)` around the stores
into xsend[] in slf_rowpush.h / x_face_send_own_row, `if (p.g.variant & 4096) return;` at the top of x_face_receive) in the
build that produced profiles/r05/xface_cost_probe.txt.  Result: every configuration lies inside the +-3 % that two
placements of the same arrays differ by (the first and the last line of each block are the SAME configuration) -- the face
buffers cost the sweep nothing measurable.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from sailfish_amd import sym
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.connector import RingExchanger
    from sailfish_amd.slab import SlabSim
    pattern = sys.argv[1] if len(sys.argv) > 1 else 'AA'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    samples = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    size = (128, 512, 512)

    class Opt(object):
        pass
    for label, variant, halo in (('periodic box (no faces)', 11, False), ('x faces: sends + receives', 11, True),
                                 ('x faces: no sends', 11 + 2048, True), ('x faces: no receives', 11 + 4096, True),
                                 ('x faces: neither', 11 + 6144, True), ('x faces: sends + receives (again)', 11, True)):
        os.environ['SLF_VARIANT'] = str(variant)
        b = HIPBackend(Opt(), 0)
        sim = SlabSim(b, sym.D3Q19, size, rank=0, world=1, access_pattern=pattern, axis='x', force_halo=halo,
                      exchanger=RingExchanger(0, 1) if halo else None, tune_placement=False)
        sim.init_synthetic()
        run = sim.step_sweep_only if halo else sim.step
        for _ in range(20):
            run()
        sim.sync()
        ms = []
        for _ in range(samples):
            e0 = b.make_event(sim.calc_stream, timing=True)
            for _ in range(steps):
                run()
            e1 = b.make_event(sim.calc_stream, timing=True)
            e1.synchronize()
            ms.append(e1.time_since(e0) / steps)
        print('%-36s %s  median %.4f ms  min %.4f  max %.4f' % (label, pattern, float(np.median(ms)), min(ms), max(ms)), flush=True)
        sim.release()


if __name__ == '__main__':
    main()
