#!/usr/bin/env python3
"""Weak scaling, 3D single-phase fluid: the lid-driven cavity, one edge^3 block per GPU, blocks stacked
along z (counterpart of the reference's benchmark/scaling/weak_single_3d.py, sized for an MI355X).

    python benchmark/scaling/weak_single_3d.py --num_blocks 1                    # one block, one process
    python benchmark/scaling/weak_single_3d.py --num_blocks 8 --gpus 0 1 2 3 4 5 6 7
                                        # the controller starts one process per GPU itself (sailfish_amd/launch.py)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        benchmark/scaling/weak_single_3d.py                                      # or: ranks started by torchrun
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))

from benchmark.scaling import util  # noqa: E402
from examples.ldc_3d import CavitySim  # noqa: E402
from sailfish.controller import LBSimulationController  # noqa: E402
from sailfish.geo import EqualSubdomainsGeometry3D  # noqa: E402


def run_benchmark(num_blocks, edge=512):
    settings = {
        'max_iters': 1500,
        'benchmark_sample_from': 500,
        'quiet': True,
        'subdomains': num_blocks,
        'conn_axis': 'z',
        'mode': 'benchmark',
        'access_pattern': 'AA',
        'lat_nx': edge,
        'lat_ny': edge,
        'lat_nz': edge * num_blocks,
    }
    ctrl = LBSimulationController(CavitySim, EqualSubdomainsGeometry3D, settings)
    timing_infos, min_timings, max_timings, subdomains = ctrl.run()
    return util.save_result('weak_3d_single', num_blocks, timing_infos, min_timings, max_timings, subdomains)


if __name__ == '__main__':
    args = util.process_cmdline()
    res = run_benchmark(args.num_blocks, args.edge)
    if res:
        print('weak_3d_single blocks=%d  MLUPS eff=%.2f comp=%.2f' % ((args.num_blocks,) + res))
