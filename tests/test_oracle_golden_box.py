"""The box-level fixture probes of tests/test_gpu_golden.py (uniform box = post-collision state after one step; single
boundary nodes collected from their neighbours) run against the CPU oracle: pins the oracle's sweep as a whole -- node
map decode, parameter table, streaming, stored rho / u -- to the reference-derived values, and keeps the probe logic
itself tested where there is no GPU."""
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


@pytest.mark.parametrize('model', ['bgk', 'mrt'])
@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_oracle_composed_step(golden_dir, name, precision, model):
    probes.composed_probe(None, golden_dir, name, precision, model, box_cls=OracleProbeBox)


@pytest.mark.parametrize('precision', ['double', 'single'])
@pytest.mark.parametrize('name', ['D2Q9', 'D3Q19'])
def test_oracle_initial_conditions(golden_dir, name, precision):
    probes.init_probe(None, golden_dir, name, precision, box_cls=OracleProbeBox)
