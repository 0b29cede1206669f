#!/usr/bin/env python
"""Pulsatile plane Poiseuille flow, D2Q9: the channel of examples/poiseuille.py with a driving term that oscillates in
time (cf. sailfish's examples/poiseuille_pulsatile.py) -- a body force a(t) = a0 sin(t) (--drive=force) or a pressure
difference of the same shape between an equilibrium-density inlet and outlet (--drive=pressure, the default).  `t` is
sym.S.time = iteration x --dt_per_lattice_time_unit (default here: 0.001, one period every 6283 steps); the values are
DynamicValue expressions the host evaluates before every step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root (the `sailfish` alias)

from sailfish.controller import LBSimulationController
from sailfish.geo import LBGeometry2D
from sailfish.lb_base import LBForcedSim
from sailfish.lb_single import LBFluidSim
from sailfish.node_type import DynamicValue
from sailfish.sym import S

from examples.poiseuille import ChannelSim, ChannelSubdomain


class PulsatileSubdomain(ChannelSubdomain):
    def boundary_conditions(self, hx, hy):
        from sympy import sin
        cfg = self.config
        along, across = (hx, hy) if cfg.horizontal else (hy, hx)
        n_along = self.gx if cfg.horizontal else self.gy
        n_across = self.gy if cfg.horizontal else self.gx
        if cfg.drive == 'pressure':
            half = 0.5 * self._pressure_drop_per_node() * n_along * sin(S.time)
            inside = (across > 0) & (across < n_across - 1)
            self.set_node(inside & (along == 0), self.pressure_bc(DynamicValue(1.0 + 3.0 * half)))
            self.set_node(inside & (along == n_along - 1), self.pressure_bc(DynamicValue(1.0 - 3.0 * half)))
        self.set_node(across == 0, self.wall_bc)
        self.set_node(across == n_across - 1, self.wall_bc)


class PulsatileSim(ChannelSim):
    subdomain = PulsatileSubdomain

    @classmethod
    def update_defaults(cls, defaults):
        ChannelSim.update_defaults(defaults)
        defaults.update({'drive': 'pressure', 'dt_per_lattice_time_unit': 0.001})

    def __init__(self, config):
        # (not ChannelSim.__init__: its constant body force is replaced by the oscillating one)
        LBFluidSim.__init__(self, config)
        LBForcedSim.__init__(self, config)
        if config.drive == 'force':
            from sympy import sin
            accel = sin(S.time) * self.subdomain.max_v * 8.0 * config.visc / self.subdomain.channel_width(config) ** 2
            self.add_body_force(DynamicValue(accel, 0.0) if config.horizontal else DynamicValue(0.0, accel))


if __name__ == '__main__':
    LBSimulationController(PulsatileSim, LBGeometry2D).run()
