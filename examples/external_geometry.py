#!/usr/bin/env python
"""Flow through a geometry given as a Boolean wall map (True = solid), body-force driven, D3Q19 -- the
set-up of sailfish's examples/external_geometry.py.  Without --geometry the demo geometry of that
example is generated: a 128 x 41 x 41 pipe along x whose radius varies sinusoidally,
(z - 20)^2 + (y - 20)^2 > (19.3 (0.8 + 0.2 sin(2 pi x / 128)))^2  ->  wall.

With --node_addressing=indirect only the fluid nodes and one layer of wall / ghost nodes around them own
storage for their distributions (load_active_node_map)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from sailfish.controller import LBSimulationController
from sailfish.lb_base import LBForcedSim
from sailfish.lb_single import LBFluidSim
from sailfish.node_type import NTFullBBWall
from sailfish.subdomain import Subdomain3D


def wavy_pipe(nx=128, ny=41, nz=41):
    hz, hy, hx = np.mgrid[0:nz, 0:ny, 0:nx]
    radius = 0.5 * (min(ny, nz) - 2.4) * (0.8 + 0.2 * np.sin(2 * np.pi * hx / float(nx)))
    return (hz - (nz - 1) / 2.0) ** 2 + (hy - (ny - 1) / 2.0) ** 2 > radius ** 2


class GeometrySubdomain(Subdomain3D):
    def initial_conditions(self, sim, hx, hy, hz):
        sim.rho[:] = 1.0

    def _walls(self, hx, hy, hz):
        return self.select_subdomain(self.config._wall_map, hx, hy, hz)

    def boundary_conditions(self, hx, hy, hz):
        self.set_node(self._walls(hx, hy, hz), NTFullBBWall)

    def load_active_node_map(self, hx, hy, hz):       # --node_addressing=indirect only
        self.set_active_node_map_from_wall_map(self._walls(hx, hy, hz))


class GeometrySim(LBFluidSim, LBForcedSim):
    subdomain = GeometrySubdomain

    @classmethod
    def add_options(cls, group, dim):
        group.add_argument('--geometry', type=str, default='',
                           help='.npy file with a Boolean [nz, ny, nx] array, True = wall (default: wavy pipe)')

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'visc': 0.01, 'grid': 'D3Q19', 'periodic_x': True,
                         'lat_nx': 128, 'lat_ny': 41, 'lat_nz': 41})

    @classmethod
    def modify_config(cls, config):
        if getattr(config, 'geometry', ''):
            wall_map = np.load(config.geometry).astype(bool)
            config.lat_nz, config.lat_ny, config.lat_nx = wall_map.shape
        else:
            wall_map = wavy_pipe(config.lat_nx, config.lat_ny, config.lat_nz)
        # one layer for the ghost nodes (envelope size 1); solid, unless the axis is periodic
        config._wall_map = np.pad(wall_map, 1, 'constant', constant_values=True)

    def __init__(self, config):
        super(GeometrySim, self).__init__(config)
        self.add_body_force((1e-5, 0.0, 0.0))


if __name__ == '__main__':
    LBSimulationController(GeometrySim).run()
