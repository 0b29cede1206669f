#!/usr/bin/env python
"""Liquid-vapour phase separation in the single-component Shan-Chen model, D2Q9, classic
pseudopotential psi = 1 - exp(-rho) (the set-up of sailfish's examples/sc_phase_separation.py:
rho = 0.693 + U[0, 0.01), G = -5, nu = 1/6, fully periodic 256 x 256)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root (the `sailfish` alias)

import numpy as np

from sailfish.controller import LBSimulationController
from sailfish.geo import LBGeometry2D
from sailfish.lb_single import LBSingleFluidShanChen
from sailfish.subdomain import Subdomain2D


class VapourSubdomain(Subdomain2D):
    def boundary_conditions(self, hx, hy):
        pass

    def initial_conditions(self, sim, hx, hy):
        rng = np.random.RandomState(self.config.seed)
        sim.rho[:] = 0.693 + rng.rand(*sim.rho.shape) / 100


class PhaseSeparationSim(LBSingleFluidShanChen):
    subdomain = VapourSubdomain

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'lat_nx': 256, 'lat_ny': 256, 'G': -5.0, 'visc': 1.0 / 6.0, 'periodic_x': True,
                         'periodic_y': True, 'sc_potential': 'classic', 'every': 20})


if __name__ == '__main__':
    LBSimulationController(PhaseSeparationSim, LBGeometry2D).run()
