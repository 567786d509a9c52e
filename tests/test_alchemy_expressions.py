"""The soft-core energy expressions pinned to the reference's OWN string literals.

tests/golden/reference_alchemy_expressions.json was produced by tests/golden/make_golden_alchemy_strings.py: the bodies of
AbsoluteAlchemicalFactory._get_sterics_energy_expressions / _get_electrostatics_energy_expressions / _get_reaction_field_unique_expression /
_get_pme_direct_space_unique_expression (/root/reference/openmmtools/alchemy/alchemy.py:1356-1537) taken out of the syntax tree and
executed, and the resulting expressions evaluated on a grid.  Here

* the literals this package writes into a store (openmmtools_amd/_alchemical_xml.py) must be those strings, character for character;
* the f64 oracle (oracle/forcefield.py), the C++ port (libremd_cpu.so, through the C ABI) and -- under -m gpu -- the HIP soft-core
  kernels must reproduce the VALUES pair by pair: one alchemical + one plain particle, as a nonbonded pair and as a 1-4 exception.
"""
import json
import os

import numpy as np
import pytest

import oracle
from oracle.forcefield import ForceFieldOracle
from openmmtools_amd import alchemy, _alchemical_xml as ax
from openmmtools_amd.system import System, NonbondedForce, system_to_desc
from openmmtools_amd._engine import HipEngine

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, 'golden', 'reference_alchemy_expressions.json')))
E = G['expressions']
KB = 0.008314462618153242
CPU_LIB = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')
LAMBDAS = [1.0, 0.7, 0.35, 0.0]
L = 4.0


def test_the_written_literals_are_the_references():
    assert ax.sterics_exception_expression() == E['sterics_exception']
    assert ax._MIX_STERICS == E['sterics_mixing_rules']
    assert ax.sterics_exception_expression() + ax._MIX_STERICS == E['sterics_pair']
    assert ax.sterics_exception_expression('lambda_sterics_zero*lambda_sterics_one') == E['sterics_exception_two_regions']
    nb = NonbondedForce()
    nb.setNonbondedMethod(NonbondedForce.NoCutoff)
    assert ax.electrostatics_expressions(nb) == (E['electrostatics_nocutoff'], E['electrostatics_exception_nocutoff'])
    nb = NonbondedForce()
    nb.setNonbondedMethod(NonbondedForce.CutoffPeriodic); nb.setCutoffDistance(1.0); nb.setReactionFieldDielectric(78.3)
    assert ax.electrostatics_expressions(nb) == (E['electrostatics_rf_switched'], E['electrostatics_exception_rf'])
    assert abs(ax.ONE_4PI_EPS0 / G['ONE_4PI_EPS0'] - 1) < 1e-15


def test_the_interpreter_of_the_store_test_reads_the_reference_strings_alike():
    """tests/test_alchemical_store_cpu.py interprets the documents this package writes; on the reference's strings it must give the
    values the generator stored (it is the same algorithm written twice: this catches either drifting)."""
    from test_alchemical_store_cpu import _evaluate
    for key in ('sterics_random', 'electrostatics_nocutoff', 'electrostatics_rf_switched', 'electrostatics_rf_shifted',
                'electrostatics_pme_direct_space', 'electrostatics_pme_coulomb'):
        expr = E['sterics_pair' if key == 'sterics_random' else key]
        for s in G['samples'][key]:
            v = {k: x for k, x in s.items() if k not in ('value', 'softcore', 'softcore_beta')}
            sc = dict(G['softcore'], **s.get('softcore', {}))
            if 'softcore_beta' in s:
                sc['softcore_beta'] = s['softcore_beta']
            assert np.isclose(_evaluate(expr, dict(v, **sc)), s['value'], rtol=1e-13, atol=1e-300)


def _two_particles(sigma, epsilon, exception):
    """particle 0 alchemical, particle 1 plain; Lennard-Jones only, no dispersion correction (two particles have one)."""
    s = System()
    s.addParticle(12.0); s.addParticle(12.0)
    s.setDefaultPeriodicBoxVectors([L, 0, 0], [0, L, 0], [0, 0, L])
    nb = NonbondedForce()
    nb.setNonbondedMethod(NonbondedForce.CutoffPeriodic); nb.setCutoffDistance(1.0)
    nb.setUseSwitchingFunction(True); nb.setSwitchingDistance(0.85); nb.setUseDispersionCorrection(False)
    nb.addParticle(0.0, sigma, epsilon); nb.addParticle(0.0, sigma, epsilon)
    if exception:
        nb.addException(0, 1, 0.0, sigma, epsilon)
    s.addForce(nb)
    return alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(s, alchemy.AlchemicalRegion(alchemical_atoms=[0]))


def _groups():
    """the 'sterics' grid by (sigma, epsilon): 5 systems x 5 distances x 4 lambdas"""
    out = {}
    for s in G['samples']['sterics']:
        out.setdefault((s['sigma1'], s['epsilon1']), {}).setdefault(s['r'], {})[s['lambda_sterics']] = (s['value'], s['exception_value'])
    return out


def _positions(rs):
    x = np.zeros((len(rs), 2, 3))
    x[:, 0] = [1.0, 1.3, 0.9]
    for k, r in enumerate(rs):
        d = np.array([0.6, -0.48, 0.64])                  # unit vector
        x[k, 1] = x[k, 0] + r * d
    return x


def _check_engine(make_engine, rtol):
    for exception in (False, True):
        for (sigma, epsilon), by_r in _groups().items():
            rs = sorted(by_r)
            system = _two_particles(sigma, epsilon, exception)
            x = _positions(rs)
            eng = make_engine()
            eng.set_system(system_to_desc(system))
            eng.set_states(np.full(len(LAMBDAS), 1.0 / (KB * 300.0)), np.array(LAMBDAS), np.ones(len(LAMBDAS)), None)
            eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
            eng.set_replicas(len(rs), 0, x, None, np.tile([L, L, L], (len(rs), 1)), np.zeros(len(rs), dtype=int))
            rows = np.asarray(eng.compute_energies()) * (KB * 300.0)
            eng.close()
            want = np.array([[by_r[r][lam][1 if exception else 0] for lam in LAMBDAS] for r in rs])
            # (a row is U_r + E_alch[r][l] - E_alch[r][own] in the device's arithmetic: the error scales with the row's largest term)
            # and near r = sigma the two Lennard-Jones terms (each ~ 4 epsilon) cancel: an absolute floor of that size times rtol
            bound = 5.0 * rtol * np.abs(want) + rtol * np.abs(want).max(axis=1, keepdims=True) + 8.0 * rtol * epsilon + rtol * 1e-3
            assert np.all(np.abs(rows - want) <= bound), (exception, sigma, epsilon, np.abs(rows - want).max())


def test_oracle_reproduces_the_reference_expression_values():
    for exception in (False, True):
        for (sigma, epsilon), by_r in _groups().items():
            ff = ForceFieldOracle(system_to_desc(_two_particles(sigma, epsilon, exception)))
            rs = sorted(by_r)
            x = _positions(rs)
            for k, r in enumerate(rs):
                got = ff.state_energies(x[k], np.array([L, L, L]), np.array(LAMBDAS), np.ones(len(LAMBDAS)))
                want = np.array([by_r[r][lam][1 if exception else 0] for lam in LAMBDAS])
                assert np.allclose(got, want, rtol=1e-12, atol=1e-14), (exception, sigma, epsilon, r, got, want)


def test_cpu_port_reproduces_the_reference_expression_values():
    if not os.path.exists(CPU_LIB):
        oracle.build()
    _check_engine(lambda: HipEngine(lib_path=CPU_LIB), 1e-12)


@pytest.mark.gpu
def test_hip_softcore_kernels_reproduce_the_reference_expression_values():
    _check_engine(lambda: HipEngine(), 1e-6)
