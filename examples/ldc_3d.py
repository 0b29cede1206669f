#!/usr/bin/env python
"""Lid-driven cavity, D3Q19 (the geometry of sailfish's examples/ldc_3d.py: the z = max plane is the
lid, moving in +x; every other face is a full-way bounce-back wall)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root (the `sailfish` alias)

from sailfish.controller import LBSimulationController
from sailfish.geo import EqualSubdomainsGeometry3D
from sailfish.lb_single import LBFluidSim
from sailfish.node_type import NTFullBBWall, NTRegularizedVelocity
from sailfish.subdomain import Subdomain3D


class CavitySubdomain(Subdomain3D):
    lid_velocity = 0.05

    def boundary_conditions(self, hx, hy, hz):
        walls = ((hx == 0) | (hx == self.gx - 1) | (hy == 0) | (hy == self.gy - 1) | (hz == 0))
        self.set_node(walls, NTFullBBWall)
        self.set_node((hz == self.gz - 1) & ~walls, NTRegularizedVelocity((self.lid_velocity, 0.0, 0.0)))

    def initial_conditions(self, sim, hx, hy, hz):
        sim.rho[:] = 1.0
        sim.vx[hz == self.gz - 1] = self.lid_velocity


class CavitySim(LBFluidSim):
    subdomain = CavitySubdomain

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'lat_nx': 64, 'lat_ny': 64, 'lat_nz': 64, 'grid': 'D3Q19'})


if __name__ == '__main__':
    LBSimulationController(CavitySim, EqualSubdomainsGeometry3D).run()     # --subdomains / --conn_axis; --gpus 0 1 ...: one process per GPU
