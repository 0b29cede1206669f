"""Shared set-up of the binary Shan-Chen tests."""
import numpy as np

from sailfish_amd.geo import LBGeometry2D, LBGeometry3D
from sailfish_amd.lb_binary import LBBinaryFluidShanChen
from sailfish_amd.subdomain import Subdomain2D, Subdomain3D


def make_sim(dim, seed=7, amplitude=1e-3):
    base = Subdomain2D if dim == 2 else Subdomain3D

    class Mixture(base):
        def boundary_conditions(self, *h):
            pass

        def initial_conditions(self, sim, *h):
            # noise defined on the *global* grid so that any decomposition starts from the same state
            rng = np.random.RandomState(seed)
            gshape = (self.gy, self.gx) if dim == 2 else (self.gz, self.gy, self.gx)
            where = tuple(reversed(h))
            sim.rho[:] = 1.0 + amplitude * rng.rand(*gshape)[where]
            sim.phi[:] = 1.0 + amplitude * rng.rand(*gshape)[where]

    class Sim(LBBinaryFluidShanChen):
        subdomain = Mixture

    return Sim, (LBGeometry2D if dim == 2 else LBGeometry3D)


def config(dim, size, pattern='AB', fused=True, G12=1.2, G11=0.0, G22=0.0, visc=1.0 / 6.0, tau_phi=1.0,
           precision='single', potential='linear'):
    cfg = dict(lat_nx=size[0], lat_ny=size[1], periodic_x=True, periodic_y=True, access_pattern=pattern,
               hip_fused_periodic=fused, G11=G11, G12=G12, G22=G22, visc=visc, tau_phi=tau_phi, precision=precision,
               sc_potential=potential, force_implementation='guo', grid='D2Q9' if dim == 2 else 'D3Q19')
    if dim == 3:
        cfg.update(lat_nz=size[2], periodic_z=True)
    return cfg


def make_single_sim(dim, seed=2348, rho0=0.693, amplitude=0.01):
    from sailfish_amd.lb_single import LBSingleFluidShanChen
    base = Subdomain2D if dim == 2 else Subdomain3D

    class Vapour(base):
        def boundary_conditions(self, *h):
            pass

        def initial_conditions(self, sim, *h):
            rng = np.random.RandomState(seed)
            gshape = (self.gy, self.gx) if dim == 2 else (self.gz, self.gy, self.gx)
            sim.rho[:] = rho0 + amplitude * rng.rand(*gshape)[tuple(reversed(h))]

    class Sim(LBSingleFluidShanChen):
        subdomain = Vapour

    return Sim, (LBGeometry2D if dim == 2 else LBGeometry3D)


def single_config(dim, size, pattern='AB', fused=True, G=-5.0, visc=1.0 / 6.0, precision='single',
                  potential='classic'):
    cfg = dict(lat_nx=size[0], lat_ny=size[1], periodic_x=True, periodic_y=True, access_pattern=pattern,
               hip_fused_periodic=fused, G=G, visc=visc, precision=precision, sc_potential=potential,
               force_implementation='guo', grid='D2Q9' if dim == 2 else 'D3Q19')
    if dim == 3:
        cfg.update(lat_nz=size[2], periodic_z=True)
    return cfg


def make_forced_sim(dim, a0, a1, seed=7):
    """Binary mixture with a body-force acceleration a0 on lattice 0 and a1 on lattice 1
    (reference add_body_force(..., grid=k); examples/binary_fluid/sc_poiseuille_2d.py, sc_rayleigh_taylor_2d.py)."""
    base, geo = make_sim(dim, seed=seed)

    class Forced(base):
        def __init__(self, config):
            super(Forced, self).__init__(config)
            if a0 is not None:
                self.add_body_force(tuple(a0))
            if a1 is not None:
                self.add_body_force(tuple(a1), grid=1)

    return Forced, geo


def make_wall_sim(dim, seed=11, amplitude=1e-3, wall=4):
    """Binary mixture between two solid slabs (`wall` nodes thick each, so that their inner layers are neither fluid
    nor next to fluid: inactive under --node_addressing=indirect) with a solid block in the channel."""
    from sailfish_amd.node_type import NTFullBBWall
    base, geo = make_sim(dim, seed=seed, amplitude=amplitude)
    sub = base.subdomain

    class Walled(sub):
        def _solid(self, *h):
            hy, hx = h[1], h[0]
            solid = (hy < wall) | (hy >= self.gy - wall)
            block = (hx >= 5) & (hx < 9) & (hy >= wall + 2) & (hy < wall + 6)
            if dim == 3:
                block = block & (h[2] >= 1) & (h[2] < 5)
            return solid | block

        def boundary_conditions(self, *h):
            self.set_node(self._solid(*h), NTFullBBWall)

        def load_active_node_map(self, *h):               # --node_addressing=indirect: all nodes incl. ghosts
            self.set_active_node_map_from_wall_map(self._solid(*h))

    class Sim(base):
        subdomain = Walled

    return Sim, geo


def make_single_wall_sim(dim, wall=4, **kw):
    """Single-component vapour between two solid slabs with a solid block in the channel (as make_wall_sim)."""
    from sailfish_amd.node_type import NTFullBBWall
    base, geo = make_single_sim(dim, **kw)
    sub = base.subdomain

    class Walled(sub):
        def _solid(self, *h):
            hy, hx = h[1], h[0]
            solid = (hy < wall) | (hy >= self.gy - wall)
            block = (hx >= 5) & (hx < 9) & (hy >= wall + 2) & (hy < wall + 6)
            if dim == 3:
                block = block & (h[2] >= 1) & (h[2] < 5)
            return solid | block

        def boundary_conditions(self, *h):
            self.set_node(self._solid(*h), NTFullBBWall)

        def load_active_node_map(self, *h):
            self.set_active_node_map_from_wall_map(self._solid(*h))

    class Sim(base):
        subdomain = Walled

    return Sim, geo
