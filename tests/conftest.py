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


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
