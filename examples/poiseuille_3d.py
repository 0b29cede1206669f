#!/usr/bin/env python
"""Hagen-Poiseuille flow in a circular pipe, D3Q19, body-force or pressure driven; the pipe wall is a
staircase of full-way bounce-back nodes (cf. sailfish's examples/poiseuille_3d.py; same option names:
--flow_direction, --drive, --stationary)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root (the `sailfish` alias)

import numpy as np

from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
from sailfish.lb_base import LBForcedSim
from sailfish.lb_single import LBFluidSim
from sailfish.node_type import NTEquilibriumDensity, NTFullBBWall
from sailfish.subdomain import Subdomain3D

AXIS = {'x': 0, 'y': 1, 'z': 2}


class PipeSubdomain(Subdomain3D):
    max_v = 0.02
    wall_bc = NTFullBBWall

    @classmethod
    def width(cls, config):
        sizes = [config.lat_nx, config.lat_ny, config.lat_nz]
        del sizes[AXIS[config.flow_direction]]
        return min(sizes)

    @classmethod
    def channel_width(cls, config):
        return cls.width(config) - 1 - 2 * cls.wall_bc.location

    def _cross_section_radius_sq(self, hx, hy, hz):
        """Squared distance from the pipe axis (which runs through the centre of the cross-section)."""
        coords = [hx, hy, hz]
        sizes = [self.gx, self.gy, self.gz]
        a = AXIS[self.config.flow_direction]
        r2 = 0.0
        for i in range(3):
            if i != a:
                r2 = r2 + (coords[i] - (sizes[i] / 2 - 0.5)) ** 2
        return r2

    def boundary_conditions(self, hx, hy, hz):
        cfg = self.config
        wall = self._cross_section_radius_sq(hx, hy, hz) >= (self.channel_width(cfg) / 2.0) ** 2
        self.set_node(wall, self.wall_bc)
        if cfg.drive == 'pressure':
            a = AXIS[cfg.flow_direction]
            along = [hx, hy, hz][a]
            n = [self.gx, self.gy, self.gz][a]
            dp = self.max_v * 16.0 * cfg.visc * (n - 1) / self.channel_width(cfg) ** 2
            self.set_node((along == 0) & ~wall, NTEquilibriumDensity(1.0 + 1.5 * dp))
            self.set_node((along == n - 1) & ~wall, NTEquilibriumDensity(1.0 - 1.5 * dp))

    def initial_conditions(self, sim, hx, hy, hz):
        sim.rho[:] = 1.0
        cfg = self.config
        if cfg.stationary and cfg.drive == 'force':
            r2 = self._cross_section_radius_sq(hx, hy, hz)
            R2 = (self.channel_width(cfg) / 2.0) ** 2
            prof = np.where(r2 < R2, self.max_v * (1.0 - r2 / R2), 0.0)
            [sim.vx, sim.vy, sim.vz][AXIS[cfg.flow_direction]][:] = prof


class PipeSim(LBFluidSim, LBForcedSim):
    subdomain = PipeSubdomain

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'lat_nx': 64, 'lat_ny': 64, 'lat_nz': 64, 'visc': 0.1, 'grid': 'D3Q19'})

    @classmethod
    def add_options(cls, group, dim):
        group.add_argument('--flow_direction', type=str, default='x', choices=['x', 'y', 'z'])
        group.add_argument('--stationary', action='store_true', default=False)
        group.add_argument('--drive', type=str, default='force', choices=['force', 'pressure'])

    @classmethod
    def modify_config(cls, config):
        if config.drive == 'force':
            config.periodic_x = config.flow_direction == 'x'
            config.periodic_y = config.flow_direction == 'y'
            config.periodic_z = config.flow_direction == 'z'

    def __init__(self, config):
        super(PipeSim, self).__init__(config)
        if config.drive == 'force':
            accel = self.subdomain.max_v * 16.0 * config.visc / self.subdomain.channel_width(config) ** 2
            f = [0.0, 0.0, 0.0]
            f[AXIS[config.flow_direction]] = accel
            self.add_body_force(tuple(f))


if __name__ == '__main__':
    LBSimulationController(PipeSim, EqualSubdomainsGeometry3D).run()
