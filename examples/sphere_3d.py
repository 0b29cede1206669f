#!/usr/bin/env python
"""Body-force driven flow past a sphere in a square duct, D3Q19, periodic along x (cf. sailfish's examples/sphere_3d.py,
the case behind its regtest/subdomains/3d_sphere.py).  The sphere has a third of the duct's height and sits two
diameters behind the inlet plane, on the duct's axis; duct walls and sphere are full-way bounce-back nodes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root (the `sailfish` alias)

from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
from sailfish.lb_base import LBForcedSim
from sailfish.lb_single import LBFluidSim
from sailfish.node_type import NTFullBBWall
from sailfish.subdomain import Subdomain3D


class SphereSubdomain(Subdomain3D):
    wall_bc = NTFullBBWall

    def boundary_conditions(self, hx, hy, hz):
        duct = (hy == 0) | (hy == self.gy - 1) | (hz == 0) | (hz == self.gz - 1)
        self.set_node(duct, self.wall_bc)
        diameter = self.gy / 3.0
        r2 = (hx - 2.0 * diameter) ** 2 + (hy - self.gy / 2.0) ** 2 + (hz - self.gz / 2.0) ** 2
        self.set_node(r2 <= (diameter / 2.0) ** 2, self.wall_bc)

    def initial_conditions(self, sim, hx, hy, hz):
        sim.rho[:] = 1.0
        sim.vx[:] = 0.0
        sim.vy[:] = 0.0
        sim.vz[:] = 0.0


class SphereSim(LBFluidSim, LBForcedSim):
    subdomain = SphereSubdomain

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'lat_nx': 128, 'lat_ny': 64, 'lat_nz': 64, 'visc': 0.01, 'grid': 'D3Q19'})

    @classmethod
    def modify_config(cls, config):
        config.periodic_x = True

    def __init__(self, config):
        super(SphereSim, self).__init__(config)
        self.add_body_force((1e-5, 0.0, 0.0))


if __name__ == '__main__':
    LBSimulationController(SphereSim, EqualSubdomainsGeometry3D).run()
