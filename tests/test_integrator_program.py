"""The Langevin integrator against the reference's own step program.

tests/golden/integrator_program_reference.json was produced by tests/golden/make_golden_integrator_program.py: the class
LangevinIntegrator of openmmtools/integrators.py taken out of the syntax tree and EXECUTED on a stand-in for openmm.CustomIntegrator
that records its calls, and the recorded program interpreted with CustomIntegrator's semantics on a four-atom chain (bonds in force
group 0, angles in group 1, no constraints) with the noise this repository's engines draw for the same seed.  Here
  * the f64 oracle integrator (oracle/md_oracle.py OracleLangevin on oracle/forcefield.py) and
  * the C++ port (libremd_cpu.so, through the C ABI of include/remd_hip.h)
must reproduce positions and velocities after every step, the heat and shadow work the reference accumulates, and the Metropolis
decisions -- for ten splitting strings incl. g-BAOAB, multiple-time-step and Metropolized ones; under -m gpu the HIP integrator chain
(libremd_hip.so) is compared with the same fixture DIRECTLY at fp32 tolerances (round 6; measured worst deviations 8e-8 nm / 1.4e-5 nm/ps,
profiles/r06_1_integrator_program.txt).  Constraints are OpenMM's (addConstrainPositions / addConstrainVelocities), not the reference's:
the fixture has none."""
import json
import os

import numpy as np
import pytest

import oracle
from oracle import md_oracle
from oracle.forcefield import ForceFieldOracle
from openmmtools_amd.system import System, HarmonicBondForce, HarmonicAngleForce, system_to_desc
from openmmtools_amd._engine import HipEngine

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, 'golden', 'integrator_program_reference.json')))
CPU_LIB = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')
CASES = list(range(len(G['cases'])))


def _system():
    s = System()
    for m in G['masses']:
        s.addParticle(m)
    b = HarmonicBondForce()
    for i, j, r0, k in G['bonds']:
        b.addBond(int(i), int(j), r0, k)
    b.setForceGroup(0)
    a = HarmonicAngleForce()
    for i, j, l, t0, k in G['angles']:
        a.addAngle(int(i), int(j), int(l), t0, k)
    a.setForceGroup(1)
    s.addForce(b); s.addForce(a)
    return s


def test_the_recorded_programs_are_the_ones_the_docstrings_describe():
    """a look at the fixture itself: 'V R O R V' is kick / drift / OU / drift / kick with the reference's expressions"""
    c = G['cases'][0]
    assert c['splitting'] == 'V R O R V'
    per_dof = [(op[1], op[2]) for op in c['program'] if op[0] == 'per_dof']
    assert per_dof == [('sigma', 'sqrt(kT/m)'), ('v', 'v + (dt / 2) * f / m'), ('x', 'x + ((dt / 2) * v)'), ('x1', 'x'),
                       ('v', 'v + ((x - x1) / (dt / 2))'), ('v', '(a * v) + (b * sigma * gaussian)'), ('x', 'x + ((dt / 2) * v)'),
                       ('x1', 'x'), ('v', 'v + ((x - x1) / (dt / 2))'), ('v', 'v + (dt / 2) * f / m')]
    h = G['timestep']
    assert np.isclose(c['globals']['a'], np.exp(-G['collision_rate'] * h), rtol=1e-15)
    assert np.isclose(c['globals']['b'], np.sqrt(1 - np.exp(-2 * G['collision_rate'] * h)), rtol=1e-15)
    assert np.isclose(c['globals']['kT'], G['kB'] * G['temperature'], rtol=1e-15)


@pytest.mark.parametrize('k', CASES)
def test_oracle_integrator_follows_the_references_program(k):
    c = G['cases'][k]
    desc = system_to_desc(_system())
    integ = md_oracle.OracleLangevin(ForceFieldOracle(desc), c['splitting'], c['timestep'], G['collision_rate'], 1, G['seed'], cmm_frequency=0)
    measured = c['measure_heat'] or c['measure_shadow_work']
    if measured:
        integ.work = dict(heat=0.0, shadow_work=0.0, n_accepted=0, n_trials=0)
    x, v = np.array(G['x0']), np.array(G['v0'])
    kT = G['kB'] * G['temperature']
    for s, want in enumerate(c['trajectory']):
        x, v = integ.run(x, v, None, kT, G['replica'], s, first_step=0, n_steps=1)          # n_steps = 1: global step = iteration
        assert np.allclose(x, want['x'], rtol=0, atol=1e-12), (c['splitting'], s, np.abs(x - np.array(want['x'])).max())
        assert np.allclose(v, want['v'], rtol=0, atol=1e-10), (c['splitting'], s, np.abs(v - np.array(want['v'])).max())
        if c['measure_heat']:
            assert np.isclose(integ.work['heat'], want['heat'], rtol=1e-9, atol=1e-9)
        if c['measure_shadow_work']:
            assert np.isclose(integ.work['shadow_work'], want['shadow_work'], rtol=1e-9, atol=1e-9)
        if 'ntrials' in want:
            assert (integ.work['n_trials'], integ.work['n_accepted']) == (int(want['ntrials']), int(want['naccept']))


def check_engine(make_engine, k, atol_x, atol_v, rtol_work, mts=True):
    """an engine behind the C ABI against case k of the fixture (the CPU library and, under -m gpu, the device)"""
    c = G['cases'][k]
    if not mts and any(ch.isdigit() for ch in c['splitting']):
        pytest.skip('multiple-time-step splittings are not in the CPU library (it refuses them by name); the oracle and the device have them')
    eng = make_engine()
    eng.set_system(system_to_desc(_system()))
    eng.set_states(np.array([1.0 / (G['kB'] * G['temperature'])]))
    eng.set_integrator(c['splitting'], c['timestep'], G['collision_rate'], 1, False, 1e-8)
    eng.set_work_measurement(measure_heat=c['measure_heat'], measure_shadow_work=c['measure_shadow_work'])
    eng.seed(G['seed'])
    eng.set_replicas(1, 0, np.array(G['x0'])[None], np.array(G['v0'])[None], np.full((1, 3), 50.0), np.zeros(1, dtype=np.int64))
    worst = [0.0, 0.0]
    for s, want in enumerate(c['trajectory']):
        assert not eng.propagate(s).any()
        x, v = eng.get_replicas()[:2]
        worst = [max(worst[0], float(np.abs(x[0] - np.array(want['x'])).max())), max(worst[1], float(np.abs(v[0] - np.array(want['v'])).max()))]
        assert np.allclose(x[0], want['x'], rtol=0, atol=atol_x), (c['splitting'], s, worst)
        assert np.allclose(v[0], want['v'], rtol=0, atol=atol_v), (c['splitting'], s, worst)
        w = eng.get_work()
        if c['measure_heat']:
            assert np.isclose(w['heat'][0], want['heat'], rtol=rtol_work, atol=rtol_work)          # (the ABI accumulates work in 2^-24 kJ/mol fixed point)
        if c['measure_shadow_work'] and 'ntrials' not in want:
            assert np.isclose(w['shadow_work'][0], want['shadow_work'], rtol=rtol_work, atol=rtol_work)
        if 'ntrials' in want:
            assert (int(w['n_trials'][0]), int(w['n_accepted'][0])) == (int(want['ntrials']), int(want['naccept']))
    eng.close()
    return worst


@pytest.mark.parametrize('k', CASES)
def test_cpu_port_follows_the_references_program(k):
    if not os.path.exists(CPU_LIB):
        oracle.build()
    check_engine(lambda: HipEngine(lib_path=CPU_LIB), k, 1e-11, 1e-9, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('k', CASES)
def test_hip_chain_follows_the_references_program(k):
    """integrators.py:1404-1460 executed (the fixture) against csrc/integrate.hip through the C ABI: positions and velocities after EVERY
    step, heat / shadow work, Metropolis decisions.  fp32 state: 5e-7 nm on positions of order 0.1 nm, 5e-5 nm/ps on velocities of order
    1 nm/ps (measured 8e-8 / 1.4e-5), work to 2e-3 (2^-24 kJ/mol fixed point of fp32 kinetic energies)."""
    check_engine(lambda: HipEngine(), k, 5e-7, 5e-5, 2e-3)


@pytest.mark.parametrize('k', range(len(G['other_integrators'])))
def test_velocity_verlet_and_hmc_as_splittings_of_the_same_chain(k):
    """integrators.py:456-498 VelocityVerletIntegrator and :885-1010 HMCIntegrator are programs of their own in the reference; this
    package runs them as the splittings 'V R V' and 'O { (V R V)^n }' (openmmtools_amd/integrators.py).  Against the reference's
    programs executed: positions after every step, the Metropolis decisions, the energy change the test is made on; velocities after
    every ACCEPTED step.  After a rejected trajectory the reference keeps the velocities the trajectory ended with and this package
    hands back the negated start velocities (what the reference's own Langevin '}' does) -- both are discarded by the next step's draw."""
    from openmmtools_amd import integrators
    c = G['other_integrators'][k]
    mine = getattr(integrators, c['name'])(**c['kwargs'])
    desc = system_to_desc(_system())
    integ = md_oracle.OracleLangevin(ForceFieldOracle(desc), mine.splitting, float(mine.getStepSize()), float(mine._gamma), 1, G["seed"], cmm_frequency=0)
    hmc = c['name'] == 'HMCIntegrator'
    if hmc:
        integ.work = dict(heat=0.0, shadow_work=0.0, n_accepted=0, n_trials=0)
    x, v = np.array(G['x0']), np.array(G['v0'])
    kT = G['kB'] * G['temperature']
    for s, want in enumerate(c['trajectory']):
        before = dict(integ.work) if hmc else None
        x, v = integ.run(x, v, None, kT, G['replica'], s, first_step=0, n_steps=1)
        assert np.allclose(x, want['x'], rtol=0, atol=1e-12), (c['name'], s, np.abs(x - np.array(want['x'])).max())
        if not hmc or want['accept'] == 1.0:
            assert np.allclose(v, want['v'], rtol=0, atol=1e-10), (c['name'], s, np.abs(v - np.array(want['v'])).max())
        if hmc:
            assert integ.work['n_trials'] == int(want['ntrials']) and integ.work['n_accepted'] == int(want['naccept'])
            assert (integ.work['n_accepted'] - before['n_accepted']) == int(want['accept'])
