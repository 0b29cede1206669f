#!/usr/bin/env python3
"""Weak scaling, 3D binary Shan-Chen mixture in a closed box (counterpart of the reference's
benchmark/scaling/weak_binary_3d.py): two lattices, two halo exchanges per step (macroscopic fields,
then populations).  Started like weak_single_3d.py (--num_blocks N --gpus 0 .. N-1, or under torch.distributed.run)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))

from benchmark.scaling import util  # noqa: E402
from examples.binary_fluid.sc_separation_3d import SeparationSim, MixtureSubdomain  # noqa: E402
from sailfish.controller import LBSimulationController  # noqa: E402
from sailfish.geo import EqualSubdomainsGeometry3D  # noqa: E402
from sailfish.node_type import NTFullBBWall  # noqa: E402


class ClosedBox(MixtureSubdomain):
    def boundary_conditions(self, hx, hy, hz):
        shell = (hx == 0) | (hy == 0) | (hz == 0) | (hx == self.gx - 1) | (hy == self.gy - 1) | (hz == self.gz - 1)
        self.set_node(np.asarray(shell), NTFullBBWall)


class ClosedBoxSim(SeparationSim):
    # a class of its own (not an attribute patched onto SeparationSim): the ranks the controller starts import it by name
    subdomain = ClosedBox


def run_benchmark(num_blocks, edge=256):
    settings = {
        'max_iters': 700,
        'benchmark_sample_from': 200,
        'quiet': True,
        'subdomains': num_blocks,
        'conn_axis': 'z',
        'mode': 'benchmark',
        'periodic_x': False,
        'periodic_y': False,
        'periodic_z': False,
        'lat_nx': edge,
        'lat_ny': edge,
        'lat_nz': edge * num_blocks,
    }
    ctrl = LBSimulationController(ClosedBoxSim, EqualSubdomainsGeometry3D, settings)
    timing_infos, min_timings, max_timings, subdomains = ctrl.run()
    return util.save_result('weak_3d_binary', num_blocks, timing_infos, min_timings, max_timings, subdomains)


if __name__ == '__main__':
    args = util.process_cmdline()
    res = run_benchmark(args.num_blocks, args.edge if args.edge != 512 else 256)
    if res:
        print('weak_3d_binary blocks=%d  MLUPS eff=%.2f comp=%.2f' % ((args.num_blocks,) + res))
