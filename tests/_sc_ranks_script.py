#!/usr/bin/env python
"""Binary Shan-Chen mixture over --subdomains equal slabs (run by tests/test_gpu_two_ranks.py, one process per
subdomain when --gpus names several devices): the set-up of examples/binary_fluid/sc_separation_3d.py with initial
densities that are a function of the GLOBAL node position, so that the run does not depend on the partition."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
from sailfish.lb_binary import LBBinaryFluidShanChen
from sailfish.subdomain import Subdomain3D


def _noise(hx, hy, hz, salt):
    v = np.sin(hx * 12.9898 + hy * 78.233 + hz * 37.719 + salt) * 43758.5453
    return v - np.floor(v)


class MixtureSubdomain(Subdomain3D):
    def boundary_conditions(self, hx, hy, hz):
        pass

    def initial_conditions(self, sim, hx, hy, hz):
        sim.rho[:] = 1.0 + _noise(hx, hy, hz, 0.0) / 100.0
        sim.phi[:] = 1.0 + _noise(hx, hy, hz, 1.5) / 100.0


class SeparationSim(LBBinaryFluidShanChen):
    subdomain = MixtureSubdomain

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'lat_nx': 32, 'lat_ny': 24, 'lat_nz': 16, 'grid': 'D3Q19', 'G12': 1.2,
                         'visc': 1.0 / 6.0, 'periodic_x': True, 'periodic_y': True, 'periodic_z': True})


if __name__ == '__main__':
    LBSimulationController(SeparationSim, EqualSubdomainsGeometry3D).run()
