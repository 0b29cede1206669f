#!/usr/bin/env python
"""Spinodal decomposition of a binary Shan-Chen mixture in a periodic D3Q19 box (the set-up of
sailfish's examples/binary_fluid/sc_separation_3d.py: G12 = 1.2, nu = 1/6, both densities
1 + U[0, 1e-3))."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np

from sailfish.controller import LBSimulationController
from sailfish.geo import LBGeometry3D
from sailfish.lb_binary import LBBinaryFluidShanChen
from sailfish.subdomain import Subdomain3D


class MixtureSubdomain(Subdomain3D):
    def boundary_conditions(self, hx, hy, hz):
        pass

    def initial_conditions(self, sim, hx, hy, hz):
        rng = np.random.RandomState(self.config.seed)
        sim.rho[:] = 1.0 + rng.rand(*sim.rho.shape) / 1000.0
        sim.phi[:] = 1.0 + rng.rand(*sim.phi.shape) / 1000.0


class SeparationSim(LBBinaryFluidShanChen):
    subdomain = MixtureSubdomain

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'lat_nx': 192, 'lat_ny': 192, 'lat_nz': 192, 'grid': 'D3Q19', 'G12': 1.2,
                         'visc': 1.0 / 6.0, 'periodic_x': True, 'periodic_y': True, 'periodic_z': True})


if __name__ == '__main__':
    LBSimulationController(SeparationSim, LBGeometry3D).run()
