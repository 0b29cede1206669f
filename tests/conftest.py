import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# The oracle's OpenMP loops run over tiny test domains: with one thread per core of a 100+ core GPU host the
# fork/join cost dominates (tests took 10-40 s each).  bench.py's cpu_baseline is not affected (no conftest).
os.environ.setdefault('OMP_NUM_THREADS', '8')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


def _gpu_present():
    """A usable HIP device, asked of the product's own library (no torch import: the CPU suite stays light)."""
    try:
        from sailfish_amd.backend_hip import HIPBackend
        return HIPBackend.devices_count() > 0
    except Exception:  # noqa: BLE001 -- library missing / no driver: no GPU as far as the tests are concerned
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a machine without a GPU skips the @pytest.mark.gpu tests instead of failing them one by
    one (`-m gpu` on the GPU box and `-m "not gpu"` here are unaffected)."""
    gpu_items = [it for it in items if it.get_closest_marker('gpu')]
    expr = getattr(config.option, 'markexpr', '') or ''
    if 'gpu' in expr and 'not gpu' not in expr:
        return              # asked for explicitly (the GPU box): a missing device or library must fail loudly, not skip
    if not gpu_items or _gpu_present():
        return
    skip = pytest.mark.skip(reason='needs a real MI355X (no HIP device here)')
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
