#!/usr/bin/env python
"""Lid-driven cavity, D2Q9 (same set-up as sailfish's examples/ldc_2d.py: full-way bounce-back
walls on three sides, a regularized-velocity lid moving in +x on the top row).

    python examples/ldc_2d.py --lat_nx=256 --lat_ny=256 --visc=0.0254 --max_iters=10000 --every=1000 \\
        --output=/tmp/ldc
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root (the `sailfish` alias)

from sailfish.controller import LBSimulationController
from sailfish.lb_single import LBFluidSim
from sailfish.node_type import NTFullBBWall, NTRegularizedVelocity
from sailfish.subdomain import Subdomain2D


class CavitySubdomain(Subdomain2D):
    lid_velocity = 0.1

    def boundary_conditions(self, hx, hy):
        top = hy == self.gy - 1
        side_or_bottom = (hx == 0) | (hx == self.gx - 1) | (hy == 0)
        lid = top & ~side_or_bottom
        self.set_node(lid, NTRegularizedVelocity((self.lid_velocity, 0.0)))
        self.set_node(side_or_bottom, NTFullBBWall)

    def initial_conditions(self, sim, hx, hy):
        sim.rho[:] = 1.0
        sim.vx[hy == self.gy - 1] = self.lid_velocity


class CavitySim(LBFluidSim):
    subdomain = CavitySubdomain

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'lat_nx': 256, 'lat_ny': 256})


if __name__ == '__main__':
    LBSimulationController(CavitySim).run()
