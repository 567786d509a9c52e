"""The benchmark workloads are `testsystems.X()` with default arguments (BASELINE.json configs 1-5): the defaults of this package's test
systems against the reference's own `__init__` signatures (tests/golden/make_golden_testsystem_defaults.py took them out of the syntax
tree of openmmtools/testsystems.py and evaluated them in the MD unit system)."""
import inspect
import json
import os

import numpy as np
import pytest

from openmmtools_amd import testsystems

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, 'golden', 'reference_testsystem_defaults.json')))


def _same(a, b):
    if isinstance(a, float) or isinstance(b, float):
        return a is not None and b is not None and np.isclose(float(a), float(b), rtol=1e-12, atol=0.0)
    return a == b


def test_module_constants():
    for name, c in G['constants'].items():
        assert _same(getattr(testsystems, name), c['value']), (name, getattr(testsystems, name), c)


@pytest.mark.parametrize('cls', sorted(G['classes']))
def test_constructor_defaults_are_the_references(cls):
    ref = G['classes'][cls]
    sig = inspect.signature(getattr(testsystems, cls).__init__)
    for name, d in ref['arguments'].items():
        assert name in sig.parameters, (cls, name, 'missing from the signature')
        if 'value' in d:
            assert _same(sig.parameters[name].default, d['value']), (cls, name, sig.parameters[name].default, d)
    # what HostGuestExplicit hands to createSystem through its inner dictionary: this package has them as arguments
    spell = dict(rigidWater='rigid_water')
    for name, d in ref.get('create_system_defaults', {}).items():
        if 'value' in d:
            mine = spell.get(name, name)
            assert mine in sig.parameters and _same(sig.parameters[mine].default, d['value']), (cls, name, d)


def test_default_sizes_of_the_benchmark_systems():
    """what the defaults build (BASELINE.json names the atom counts): 2269 / 4491 (README of the reference's data) / 23558 atoms"""
    assert testsystems.AlanineDipeptideExplicit().system.getNumParticles() == 2269
    assert testsystems.DHFRExplicit().system.getNumParticles() == 23558
    lj = testsystems.LennardJonesFluid(nparticles=512)
    nb = [f for f in lj.system.getForces() if type(f).__name__ == 'NonbondedForce'][0]
    q, sigma, eps = nb.getParticleParameters(0)
    ref = G['classes']['LennardJonesFluid']['arguments']
    assert q == 0.0 and np.isclose(sigma, ref['sigma']['value'], rtol=1e-12) and np.isclose(eps, ref['epsilon']['value'], rtol=1e-12)
    # testsystems.py: cutoff = 3 sigma by default, switching starts switch_width inside it
    assert np.isclose(nb.getCutoffDistance(), 3.0 * ref['sigma']['value'], rtol=1e-12)
