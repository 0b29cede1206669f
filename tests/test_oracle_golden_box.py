"""The box-level fixture probes of tests/test_gpu_golden.py (uniform box = post-collision state after one step; single
boundary nodes collected from their neighbours) run against the CPU oracle: pins the oracle's sweep as a whole -- node
map decode, parameter table, streaming, stored rho / u -- to the reference-derived values, and keeps the probe logic
itself tested where there is no GPU."""
import os

import numpy as np
import pytest

from tests import test_gpu_golden as probes
from tests._oracle_box import OracleBox


class OracleProbeBox(OracleBox):
    """BoxSim's probe interface on top of the oracle twin."""

    def __init__(self, backend, desc, periodic=(False, False, False), node_map=None):
        OracleBox.__init__(self, desc, periodic, node_map)

    def set_dist(self, host, which):
        self.dist[which][...] = np.asarray(host, dtype=self.dtype).reshape(self.dist[which].shape)

    def get_dist(self):
        return self.current_dist().reshape((self.Q,) + self.shape)

    def fetch_fields(self):
        return self.rho, self.v

    def sync(self):
        pass

    def release(self):
        pass


@pytest.mark.parametrize('fused', [1, 0], ids=['in_sweep_wrap', 'ghost_pbc'])
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_oracle_sweep_collisions(golden_dir, name, precision, fused):
    probes.collision_probe(None, golden_dir, name, precision, fused, box_cls=OracleProbeBox)


@pytest.mark.parametrize('kind', sorted(probes.BC_CASES))
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_oracle_sweep_boundary_nodes(golden_dir, name, precision, kind):
    probes.boundary_probe(None, golden_dir, name, precision, kind, box_cls=OracleProbeBox)


@pytest.mark.parametrize('fused', [1, 0], ids=['in_sweep_wrap', 'ghost_pbc'])
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_oracle_minimize_roundoff(golden_dir, name, precision, fused):
    probes.roundoff_probe(None, golden_dir, name, precision, fused, box_cls=OracleProbeBox)


@pytest.mark.parametrize('kind', sorted(probes.RO_BC_CASES))
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_oracle_minimize_roundoff_boundary_nodes(golden_dir, name, precision, kind):
    probes.roundoff_boundary_probe(None, golden_dir, name, precision, kind, box_cls=OracleProbeBox)


@pytest.mark.parametrize('model', ['bgk', 'mrt'])
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_oracle_composed_step(golden_dir, name, precision, model):
    probes.composed_probe(None, golden_dir, name, precision, model, box_cls=OracleProbeBox)


@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_oracle_initial_conditions(golden_dir, name, precision):
    probes.init_probe(None, golden_dir, name, precision, box_cls=OracleProbeBox)


def test_reference_regularized_node_is_inconsistent_under_minimize_roundoff(golden_dir):
    """Why the regularized (and Zou-He) nodes are refused under --minimize_roundoff: applied to a node that is exactly
    at equilibrium the reference's own expressions return the equilibrium in the standard formulation (defect 0) and
    something O(1) away from it under the option -- ex_flux adds c_s^2 for the shifted populations (sym.py:684-695) while
    ex_eq_flux keeps multiplying by the density delta (sym.py:697-703).  Evaluated from the reference's sympy objects
    (tools/capture_goldens.py: roundoff_bc_goldens)."""
    for name in ('D2Q9', 'D3Q19'):
        d = np.load(os.path.join(str(golden_dir), 'arith_ro_bc_%s.npz' % name))['regvel_defect_std_vs_roundoff']
        assert np.all(d[:, 0] < 1e-12) and np.all(d[:, 1] > 0.1)
