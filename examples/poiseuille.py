#!/usr/bin/env python
"""Plane Poiseuille flow, D2Q9: body-force driven (periodic along the flow) or pressure driven
(equilibrium-density inlet / outlet), full-way or half-way bounce-back walls.  Option names follow
sailfish's examples/poiseuille.py (--horizontal, --drive, --wall, --stationary)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root (the `sailfish` alias)

import numpy as np

from sailfish.controller import LBSimulationController
from sailfish.geo import LBGeometry2D
from sailfish.lb_base import LBForcedSim
from sailfish.lb_single import LBFluidSim
from sailfish.node_type import NTEquilibriumDensity, NTFullBBWall, NTHalfBBWall
from sailfish.subdomain import Subdomain2D


class ChannelSubdomain(Subdomain2D):
    max_v = 0.02
    wall_bc = NTFullBBWall
    pressure_bc = NTEquilibriumDensity      # any density node type: NTZouHeDensity, NTRegularizedDensity

    @classmethod
    def width(cls, config):
        return config.lat_ny if config.horizontal else config.lat_nx

    @classmethod
    def channel_width(cls, config):
        # distance between the two effective wall positions
        return cls.width(config) - 1 - 2 * cls.wall_bc.location

    @classmethod
    def velocity_profile(cls, config, across):
        """Analytic parabola at the coordinate `across` (node index across the channel)."""
        w = cls.channel_width(config)
        d = across - cls.wall_bc.location
        return 4.0 * cls.max_v / w ** 2 * d * (w - d)

    def _pressure_drop_per_node(self):
        return self.max_v * 8.0 * self.config.visc / self.channel_width(self.config) ** 2

    def boundary_conditions(self, hx, hy):
        cfg = self.config
        along, across = (hx, hy) if cfg.horizontal else (hy, hx)
        n_along = self.gx if cfg.horizontal else self.gy
        n_across = self.gy if cfg.horizontal else self.gx
        if cfg.drive == 'pressure':
            dp = self._pressure_drop_per_node() * n_along
            inside = (across > 0) & (across < n_across - 1)
            self.set_node(inside & (along == 0), self.pressure_bc(1.0 + 1.5 * dp))
            self.set_node(inside & (along == n_along - 1), self.pressure_bc(1.0 - 1.5 * dp))
        self.set_node(across == 0, self.wall_bc)
        self.set_node(across == n_across - 1, self.wall_bc)

    def initial_conditions(self, sim, hx, hy):
        cfg = self.config
        sim.rho[:] = 1.0
        if not cfg.stationary:
            return
        along, across = (hx, hy) if cfg.horizontal else (hy, hx)
        if cfg.drive == 'pressure':
            n_along = self.gx if cfg.horizontal else self.gy
            sim.rho[:] = 1.0 + 3.0 * self._pressure_drop_per_node() * (n_along / 2.0 - along)
        else:
            (sim.vx if cfg.horizontal else sim.vy)[:] = self.velocity_profile(cfg, across)


class ChannelSim(LBFluidSim, LBForcedSim):
    subdomain = ChannelSubdomain

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'lat_nx': 128, 'lat_ny': 128, 'visc': 0.1})

    @classmethod
    def add_options(cls, group, dim):
        group.add_argument('--horizontal', action='store_true', default=False, help='flow along the X axis')
        group.add_argument('--stationary', action='store_true', default=False,
                           help='start from the analytic steady state')
        group.add_argument('--drive', type=str, default='force', choices=['force', 'pressure'])
        group.add_argument('--wall', type=str, default='fullbb', choices=['fullbb', 'halfbb'])

    @classmethod
    def modify_config(cls, config):
        if config.drive == 'force':
            config.periodic_x = bool(config.horizontal)
            config.periodic_y = not config.horizontal
        cls.subdomain.wall_bc = NTHalfBBWall if config.wall == 'halfbb' else NTFullBBWall

    def __init__(self, config):
        super(ChannelSim, self).__init__(config)
        if config.drive == 'force':
            accel = self.subdomain.max_v * 8.0 * config.visc / self.subdomain.channel_width(config) ** 2
            self.add_body_force((accel, 0.0) if config.horizontal else (0.0, accel))


if __name__ == '__main__':
    LBSimulationController(ChannelSim, LBGeometry2D).run()
