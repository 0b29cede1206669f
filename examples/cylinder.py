#!/usr/bin/env python
"""Body-force driven flow past a circular cylinder between two walls, D2Q9, periodic along the flow (cf. sailfish's
examples/cylinder.py, the case behind its regtest/subdomains/2d_cylinder.py; same options: --vertical turns the
channel by 90 degrees).  The cylinder has a third of the channel's width and sits two diameters downstream of the
inlet plane; walls and cylinder are full-way bounce-back nodes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root (the `sailfish` alias)

from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry2D
from sailfish.lb_base import LBForcedSim
from sailfish.lb_single import LBFluidSim
from sailfish.node_type import NTFullBBWall
from sailfish.subdomain import Subdomain2D


class CylinderSubdomain(Subdomain2D):
    wall_bc = NTFullBBWall

    def boundary_conditions(self, hx, hy):
        along, across = (hy, hx) if self.config.vertical else (hx, hy)
        n_across = self.gx if self.config.vertical else self.gy
        self.set_node((across == 0) | (across == n_across - 1), self.wall_bc)
        diameter = n_across / 3
        centre_along, centre_across = 2 * diameter, n_across / 2
        inside = (along - centre_along) ** 2 + (across - centre_across) ** 2 < diameter ** 2 / 4.0
        self.set_node(inside, self.wall_bc)

    def initial_conditions(self, sim, hx, hy):
        sim.rho[:] = 1.0
        sim.vx[:] = 0.0
        sim.vy[:] = 0.0


class CylinderSim(LBFluidSim, LBForcedSim):
    subdomain = CylinderSubdomain
    acceleration = 1e-5

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'lat_nx': 256, 'lat_ny': 256, 'visc': 0.1})

    @classmethod
    def add_options(cls, group, dim):
        group.add_argument('--vertical', action='store_true', default=False)

    @classmethod
    def modify_config(cls, config):
        if config.vertical:
            config.periodic_y = True
        else:
            config.periodic_x = True

    def __init__(self, config):
        super(CylinderSim, self).__init__(config)
        self.add_body_force((0.0, self.acceleration) if config.vertical else (self.acceleration, 0.0))


if __name__ == '__main__':
    LBSimulationController(CylinderSim, EqualSubdomainsGeometry2D).run()
