import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu through gpurun)')
    config.addinivalue_line('markers', 'slow: long statistical test')


@pytest.fixture(scope='session')
def hip_engine_factory():
    """Creates HipEngine handles; fails loudly (no CPU fallback) when the GPU or the .so is missing."""
    from openmmtools_amd._engine import HipEngine
    made = []

    def make(**kw):
        e = HipEngine(**kw)
        made.append(e)
        return e
    yield make
    for e in made:
        e.close()
