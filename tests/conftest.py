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


@pytest.fixture(autouse=True)
def _numpy_global_stream_per_test(request):
    """A move applied outside a sampler (mcmc.MCMCSampler without a seed) seeds its engine from numpy's GLOBAL stream, as the
    reference's moves draw from OpenMM's: the CPU suite must not depend on the operating system's entropy (a barostat test with five
    attempts failed once in a few dozen runs), so every CPU test starts that stream from its own name.  GPU tests are left alone: their
    seeds could not be tried out on the hardware when this was written."""
    if request.node.get_closest_marker('gpu') is None:
        import zlib
        import numpy as np
        np.random.seed(zlib.crc32(request.node.nodeid.encode()) & 0x7fffffff)
    yield
