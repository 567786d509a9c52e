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


@pytest.mark.parametrize('cls,name', [('AlanineDipeptideExplicit', 'alanine-dipeptide-explicit'), ('HostGuestExplicit', 'cb7-b2-explicit'),
                                      ('DHFRExplicit', 'dhfr-explicit')])
def test_stored_systems_against_digests_of_the_references_amber_files(cls, name):
    """tests/golden/prmtop_digests.json: sums and counts taken from the reference's prmtop / inpcrd files by a reader of its own
    (tests/golden/make_golden_prmtop_digests.py; unit conversions calibrated by the alanine dipeptide system, which
    tests/test_openmm_fixture.py pins field by field to the System OpenMM built from the same prmtop).  The stored systems of the three
    explicit-solvent configs carry exactly these parameters."""
    D = json.load(open(os.path.join(HERE, 'golden', 'prmtop_digests.json')))[name]
    t = getattr(testsystems, cls)()
    s = t.system
    n = s.getNumParticles()
    assert n == D['n_atoms'] and s.getNumConstraints() == D['n_constraints']
    mass = np.array([s.getParticleMass(i) for i in range(n)])
    assert np.isclose(mass.sum(), D['mass_sum'], rtol=1e-12)
    forces = {type(f).__name__: f for f in s.getForces()}
    nb = forces['NonbondedForce']
    p = np.array([nb.getParticleParameters(i) for i in range(n)])
    assert abs(p[:, 0].sum() - D['charge_sum']) < 1e-6 and np.isclose(np.abs(p[:, 0]).sum(), D['charge_abs_sum'], rtol=1e-9)
    on = p[:, 2] > 0
    assert int(on.sum()) == D['n_epsilon_nonzero']
    assert np.isclose(p[on, 1].sum(), D['sigma_sum_where_epsilon_nonzero'], rtol=1e-9) and np.isclose(p[:, 2].sum(), D['epsilon_sum'], rtol=1e-9)
    assert np.isclose(p[on, 1].max(), D['sigma_max'], rtol=1e-9) and np.isclose(p[:, 2].max(), D['epsilon_max'], rtol=1e-9)
    bf, af, tf = forces['HarmonicBondForce'], forces['HarmonicAngleForce'], forces['PeriodicTorsionForce']
    bonds = np.array([bf.getBondParameters(k)[2:] for k in range(bf.getNumBonds())]).reshape(-1, 2)
    assert len(bonds) == D['n_bonds'] and np.isclose(bonds[:, 1].sum(), D['bond_k_sum'], rtol=1e-9) and np.isclose(bonds[:, 0].sum(), D['bond_r0_sum'], rtol=1e-9)
    angles = np.array([af.getAngleParameters(k)[3:] for k in range(af.getNumAngles())]).reshape(-1, 2)
    assert len(angles) == D['n_angles'] and np.isclose(angles[:, 1].sum(), D['angle_k_sum'], rtol=1e-9) and np.isclose(angles[:, 0].sum(), D['angle_theta_sum'], rtol=1e-9)
    tors = np.array([tf.getTorsionParameters(k)[4:] for k in range(tf.getNumTorsions())], dtype=float).reshape(-1, 3)
    nz = tors[:, 2] != 0.0
    assert int(nz.sum()) == D['n_dihedrals_nonzero'] and np.isclose(tors[:, 2].sum(), D['dihedral_k_sum'], rtol=1e-9)
    assert np.isclose(tors[nz, 0].sum(), D['dihedral_periodicity_sum_nonzero'], rtol=1e-12) and np.isclose(tors[nz, 1].sum(), D['dihedral_phase_sum_nonzero'], rtol=1e-9)
    exc = np.array([nb.getExceptionParameters(k)[2:] for k in range(nb.getNumExceptions())], dtype=float).reshape(-1, 3)
    live = (exc[:, 0] != 0.0) | (exc[:, 2] != 0.0)                      # the 1-4 pairs; 1-2 and 1-3 exclusions carry zeros
    assert len(exc) == D['n_exceptions'] and int(live.sum()) == D['n_14']
    assert np.isclose(exc[:, 0].sum(), D['charge_product_14_sum'], rtol=1e-9) and np.isclose(exc[:, 2].sum(), D['epsilon_14_sum'], rtol=1e-9)
    assert np.isclose(exc[exc[:, 2] > 0, 1].sum(), D['sigma_14_sum_where_epsilon_nonzero'], rtol=1e-9)
    box = np.diag(np.array(s.getDefaultPeriodicBoxVectors(), dtype=float).reshape(3, 3))
    assert np.allclose(box, D['box'], rtol=1e-7)
    x = np.asarray(t.positions, dtype=float)
    assert np.isclose(x.sum(), D['position_sum'], rtol=1e-7) and np.isclose(np.abs(x).sum(), D['position_abs_sum'], rtol=1e-7)
    assert (getattr(t, 'velocities', None) is not None) == D['has_velocities']
