"""Simulations for the runner-level tests of the do-nothing outlet and the full-slip wall.  The first is the set-up of
the reference's tests/gpu/do_nothing_node.py (full-way bounce-back walls on y, a regularized-velocity inlet, NTDoNothing
on the last column, everything moving at 0.05 to begin with); test-only."""
import sailfish  # noqa: F401  (the sailfish.* aliases)
from sailfish.lb_base import LBForcedSim
from sailfish.lb_single import LBFluidSim
from sailfish.node_type import NTDoNothing, NTFullBBWall, NTRegularizedVelocity, NTSlip
from sailfish.subdomain import Subdomain2D, Subdomain3D
from sailfish.sym import D2Q9, D3Q19


class OpenChannelSubdomain(Subdomain2D):
    u_in = 0.05

    def boundary_conditions(self, hx, hy):
        wall = (hy == 0) | (hy == self.gy - 1)
        self.set_node(wall, NTFullBBWall)
        self.set_node(~wall & (hx == 0), NTRegularizedVelocity((self.u_in, 0.0)))
        self.set_node(~wall & (hx == self.gx - 1), NTDoNothing)

    def initial_conditions(self, sim, hx, hy):
        sim.rho[:] = 1.0
        sim.vx[:] = self.u_in


class OpenChannelSim(LBFluidSim):
    subdomain = OpenChannelSubdomain


class OpenDuctSubdomain(Subdomain3D):
    u_in = 0.04

    def boundary_conditions(self, hx, hy, hz):
        wall = (hy == 0) | (hy == self.gy - 1)
        self.set_node(wall, NTFullBBWall)
        self.set_node(~wall & (hz == 0), NTRegularizedVelocity((0.0, 0.0, self.u_in)))
        self.set_node(~wall & (hz == self.gz - 1), NTDoNothing)

    def initial_conditions(self, sim, hx, hy, hz):
        sim.rho[:] = 1.0
        sim.vz[:] = self.u_in


class OpenDuctSim(LBFluidSim):
    """Flow along z between walls on y, periodic along x; the outlet is a z face."""
    subdomain = OpenDuctSubdomain

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'periodic_x': True})


class SlipChannelSubdomain(Subdomain2D):
    """Flow along x between two full-slip walls: nothing holds the fluid back.  `u0`: the uniform initial velocity."""
    u0 = 0.0

    def boundary_conditions(self, hx, hy):
        self.set_node(hy == 0, NTSlip(orientation=D2Q9.vec_to_dir([0, 1])))
        self.set_node(hy == self.gy - 1, NTSlip(orientation=D2Q9.vec_to_dir([0, -1])))

    def initial_conditions(self, sim, hx, hy):
        sim.rho[:] = 1.0
        sim.vx[:] = self.u0


class SlipChannelSim(LBFluidSim, LBForcedSim):
    subdomain = SlipChannelSubdomain
    accel = 1e-5

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'periodic_x': True})

    def __init__(self, config):
        super(SlipChannelSim, self).__init__(config)
        if self.accel:
            self.add_body_force((self.accel, 0.0))


class SlipDuctSubdomain(Subdomain3D):
    def boundary_conditions(self, hx, hy, hz):
        self.set_node(hy == 0, NTSlip(orientation=D3Q19.vec_to_dir([0, 1, 0])))
        self.set_node(hy == self.gy - 1, NTSlip(orientation=D3Q19.vec_to_dir([0, -1, 0])))

    def initial_conditions(self, sim, hx, hy, hz):
        sim.rho[:] = 1.0


class SlipDuctSim(LBFluidSim, LBForcedSim):
    subdomain = SlipDuctSubdomain
    accel = 1e-5

    @classmethod
    def update_defaults(cls, defaults):
        defaults.update({'periodic_x': True, 'periodic_z': True})

    def __init__(self, config):
        super(SlipDuctSim, self).__init__(config)
        self.add_body_force((self.accel, 0.0, 0.0))
