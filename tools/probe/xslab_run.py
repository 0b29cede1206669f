"""The single-rank RCCL x-slab of BASELINE config 4 (128 x 512 x 512 through SlabSim) for a kernel trace:
    rocprofv3 --kernel-trace ... -- python -m torch.distributed.run --nproc-per-node 1 ... tools/probe/xslab_run.py AA 4 [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from sailfish_amd import sym
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.connector import init_distributed
    from sailfish_amd.slab import SlabSim
    pattern, k = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    axis = sys.argv[4] if len(sys.argv) > 4 else 'x'
    size = (128, 512, 512) if axis == 'x' else (512, 512, 128)
    torch.cuda.set_device(0)

    class Opt(object):
        pass
    backend = HIPBackend(Opt(), 0)
    init_distributed(force=True)
    os.environ['SLF_XFACE_CHUNKS'] = k
    sim = SlabSim(backend, sym.D3Q19, size, rank=0, world=1, access_pattern=pattern, axis=axis, force_halo=True,
                  tune_placement=False)
    sim.init_synthetic()
    for _ in range(20):
        sim.step()
    sim.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
    sim.sync()
    print('xslab_run %s K=%s axis=%s: %.4f ms per step' % (pattern, k, axis, (time.perf_counter() - t0) / steps * 1e3))
    sim.release()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
