"""A simulation whose subdomain 1 fails while it sets up its initial state (tests/test_launch.py)."""
from examples.ldc_3d import CavitySim, CavitySubdomain


class FailingSubdomain(CavitySubdomain):
    def initial_conditions(self, sim, hx, hy, hz):
        if self.spec.id == 1:
            raise ValueError('subdomain 1 fails on purpose')
        CavitySubdomain.initial_conditions(self, sim, hx, hy, hz)


class FailingSim(CavitySim):
    subdomain = FailingSubdomain
