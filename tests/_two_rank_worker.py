"""One rank of a real multi-process slab run (launched by torch.distributed.run from test_gpu_two_ranks.py): SlabSim
over torch.distributed, its result gathered on rank 0 and compared there with the single-slab run of the whole box.
usage: _two_rank_worker.py AXIS PATTERN MODEL"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    axis, pattern, model = sys.argv[1:4]
    import torch
    import torch.distributed as dist
    from sailfish_amd import sym
    from sailfish_amd.backend_hip import HIPBackend
    from sailfish_amd.connector import init_distributed
    from sailfish_amd.slab import AXES, SlabSim

    class Opt(object):
        pass
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dev = int(os.environ.get('SLF_FORCE_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(dev)
    init_distributed(force=True)
    n = (48, 12, 10)
    a = AXES[axis]
    sim = SlabSim(HIPBackend(Opt(), dev), sym.D3Q19, n, rank=rank, world=world, model=model, access_pattern=pattern,
                  visc=0.02, axis=axis)
    sim.init_synthetic(seed=7)
    steps = 9
    for _ in range(steps):
        sim.step()
    sim.sync()
    mine = {'dist': sim.real_view(sim.get_dist()), 'rho': sim.real_view(sim.rho),
            'v': [sim.real_view(sim.v[d]) for d in range(3)]}
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    ok = True
    if rank == 0:
        np_axis = 3 - a
        got = np.concatenate([p['dist'] for p in parts], axis=np_axis)
        whole = list(n)
        whole[a] *= world
        one = SlabSim(HIPBackend(Opt(), dev), sym.D3Q19, tuple(whole), rank=0, world=1, model=model,
                      access_pattern=pattern, visc=0.02)
        one.set_fields(np.concatenate([p['rho'] for p in parts], axis=np_axis - 1),
                       [np.concatenate([p['v'][d] for p in parts], axis=np_axis - 1) for d in range(3)])
        one.initial_conditions()
        for _ in range(steps):
            one.step()
        ok = bool(np.array_equal(got, one.real_view(one.get_dist())))
        print('TWO_RANK_PARITY %s backend=%s world=%d' % ('OK' if ok else 'MISMATCH', dist.get_backend(), world))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
