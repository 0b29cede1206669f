"""Where does the host spend its time while it enqueues the chunked x-slab step?  

Runs the single-rank RCCL x-slab of BASELINE config 4 (128 x 512 x 512) through SlabSim and wraps every C-ABI call
category of the step loop with perf_counter: kernel launches, event records, stream waits, RCCL batches.  Prints the
per-category host time per step, and the per-step total for a few windows of the run (is the host slow all the time,
or only once some queue is full?).

    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/probe/host_time_probe.py [K ...]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from sailfish_amd import sym
    from sailfish_amd.backend_hip import HIPBackend, HIPEvent, HIPStream
    from sailfish_amd.connector import init_distributed
    from sailfish_amd.slab import SlabSim

    torch.cuda.set_device(0)

    class Opt(object):
        pass
    backend = HIPBackend(Opt(), 0)
    init_distributed(force=True)
    acc = {}

    def timed(name, fn):
        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        return w

    backend.run_kernel = timed('launch', backend.run_kernel)
    HIPEvent.record = timed('record', HIPEvent.record)
    HIPStream.wait_for_event = timed('wait', HIPStream.wait_for_event)
    from sailfish_amd.backend_hip import HIPPlan
    HIPPlan.run = timed('plan_run', HIPPlan.run)
    ks = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
    steps = int(os.environ.get('PROBE_STEPS', '400'))
    for pattern in ('AA', 'AB'):
        for k in ks:
            os.environ['SLF_XFACE_CHUNKS'] = str(k)
            sim = SlabSim(backend, sym.D3Q19, (128, 512, 512), rank=0, world=1, access_pattern=pattern, axis='x',
                          force_halo=True, tune_placement=False)
            if hasattr(sim.exchanger, "rccl"):
                sim.exchanger.rccl.run = timed("rccl", sim.exchanger.rccl.run)
            sim.init_synthetic()
            for _ in range(40):
                sim.step()
            sim.sync()
            acc.clear()
            per_step = []
            t00 = time.perf_counter()
            for _ in range(steps):
                t0 = time.perf_counter()
                sim.step()
                per_step.append(time.perf_counter() - t0)
            t_enq = time.perf_counter() - t00
            sim.sync()
            t_all = time.perf_counter() - t00
            ps = np.array(per_step) * 1e3
            out = {'pattern': pattern, 'K': len(sim.chunks.order), 'steps': steps, 'step_ms': round(t_all / steps * 1e3, 4),
                   'host_ms': round(t_enq / steps * 1e3, 4),
                   'per_category_ms_per_step': dict((n, round(v / steps * 1e3, 4)) for n, v in sorted(acc.items())),
                   'host_ms_steps_0_20': round(float(ps[:20].mean()), 4), 'host_ms_steps_20_60': round(float(ps[20:60].mean()), 4),
                   'host_ms_steps_100_200': round(float(ps[100:200].mean()), 4), 'host_ms_last_100': round(float(ps[-100:].mean()), 4),
                   'host_ms_max': round(float(ps.max()), 3), 'host_ms_median': round(float(np.median(ps)), 4)}
            print(json.dumps(out), flush=True)
            sim.release()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
