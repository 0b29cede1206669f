#!/usr/bin/env python
"""Womersley flow: the pipe of examples/poiseuille_3d.py driven by a pressure difference that oscillates in time,
dP(t) = dP0 sin(omega t) (cf. sailfish's examples/womersley.py).  The inlet and outlet are equilibrium-density nodes
whose density is a DynamicValue of sym.S.time -- evaluated on the host before every step (sailfish_amd/node_type.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root (the `sailfish` alias)

from math import sqrt

import numpy as np

from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
from sailfish.node_type import DynamicValue, NTEquilibriumDensity
from sailfish.sym import S

from examples.poiseuille_3d import AXIS, PipeSim, PipeSubdomain


class WomersleySubdomain(PipeSubdomain):
    max_v = 0.04

    def pressure_delta(self):
        """Density difference per unit length that drives a Poiseuille flow of peak velocity max_v (rho = 3 p)."""
        cfg = self.config
        return self.max_v * 16.0 * cfg.visc / self.channel_width(cfg) ** 2

    def boundary_conditions(self, hx, hy, hz):
        from sympy import sin
        cfg = self.config
        wall = self._cross_section_radius_sq(hx, hy, hz) >= (self.channel_width(cfg) / 2.0) ** 2
        self.set_node(wall, self.wall_bc)
        a = AXIS[cfg.flow_direction]
        along = [hx, hy, hz][a]
        n = [self.gx, self.gy, self.gz][a]
        half = 0.5 * self.pressure_delta() * (n - 1) * sin(S.time * cfg.omega)
        self.set_node((along == 0) & ~wall, NTEquilibriumDensity(DynamicValue(1.0 + 3.0 * half)))
        self.set_node((along == n - 1) & ~wall, NTEquilibriumDensity(DynamicValue(1.0 - 3.0 * half)))
        R = self.channel_width(cfg) / 2.0
        cfg.logger.info('Wo = %.2f, T = %.1f steps' % (R * sqrt(cfg.omega / cfg.visc), 2 * np.pi / cfg.omega))

    def womersley_profile(self, r, t):
        """Axial velocity at the normalised radii r at time t for dP(t) = dP0 sin(omega t) (real part of the classical
        solution with Bessel functions of complex argument)."""
        from scipy.special import jv
        cfg = self.config
        alpha = self.channel_width(cfg) / 2.0 * sqrt(cfg.omega / cfg.visc)
        g = self.pressure_delta()          # density gradient / 3 = pressure gradient per unit density
        z = 1j ** 1.5 * alpha
        return np.real((1 - jv(0, z * r) / jv(0, z)) * np.exp(1j * cfg.omega * t) * (1j / cfg.omega) * 1j) * g


class WomersleySim(PipeSim):
    subdomain = WomersleySubdomain

    @classmethod
    def update_defaults(cls, defaults):
        PipeSim.update_defaults(defaults)
        defaults.update({'visc': 0.01, 'drive': 'pressure', 'lat_nx': 128, 'lat_ny': 34, 'lat_nz': 34})

    @classmethod
    def add_options(cls, group, dim):          # (the controller collects the options of every class of the hierarchy)
        group.add_argument('--omega', type=float, default=0.0005, help='angular frequency of the pressure oscillation')


if __name__ == '__main__':
    LBSimulationController(WomersleySim, EqualSubdomainsGeometry3D).run()
