"""Multiple-time-step splittings of LangevinIntegrator ("V0 V1 R R O R R V1 R R O R R V1 V0", integrators.py:1036-1053,
1425-1442, 1507-1537): a V<g> substep kicks with the forces of force group g (Force.setForceGroup /
NonbondedForce.setReciprocalSpaceForceGroup) and dt / (number of V<g> in the splitting).  Host parsing, the f64 oracle and
the device (remd_set_force_groups + per-group force arrays, integrate.hip) are checked against each other and against one
exact property: when every group is named the same number of times the scheme IS the single-time-step one."""
import copy
import numpy as np
import pytest
from openmmtools_amd import testsystems as ts, integrators, unit
from openmmtools_amd.system import system_to_desc, NonbondedForce, HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce
from oracle.forcefield import ForceFieldOracle
from oracle_engine import OracleEngine

KB = 0.008314462618153242
SEED = 0xC0FFEE
SOLVENT_SOLUTE = 'V0 V1 R R O R R V1 R R O R R V1 V0'           # integrators.py:1053


def _grouped(testsystem, bonded=1, nonbonded=0, reciprocal=None):
    system = copy.deepcopy(testsystem.system)
    for f in system.getForces():
        if isinstance(f, (HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce)):
            f.setForceGroup(bonded)
        elif isinstance(f, NonbondedForce):
            f.setForceGroup(nonbonded)
            if reciprocal is not None:
                f.setReciprocalSpaceForceGroup(reciprocal)
    return system


def _setup(eng, system, positions, splitting, n_steps, dt=0.002, R=1):
    desc = system_to_desc(system)
    eng.set_system(desc)
    eng.set_states(np.full(R, 1.0 / (KB * 300.0)))
    eng.set_integrator(splitting, dt, 1.0, n_steps, True, 1e-8)
    eng.seed(SEED)
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (R, 1))
    eng.set_replicas(R, 0, np.tile(positions, (R, 1, 1)), None, box, np.arange(R))
    return desc


def test_host_parsing_follows_the_reference():
    li = integrators.LangevinIntegrator(splitting=SOLVENT_SOLUTE)
    assert li._mts and li._force_group_nV == {'0': 2, '1': 3}                         # integrators.py:1524-1533
    single = integrators.LangevinIntegrator(splitting='V0 R O R V0')                    # one group: all forces, dt / n_V (:1535)
    assert not single._mts and single._force_group_nV == {'0': 2}
    with pytest.raises(AssertionError):
        integrators.LangevinIntegrator(splitting='V0 V R O R V1')                       # :1527-1529: every V names its group (an assert)
    with pytest.raises(ValueError):
        integrators.LangevinIntegrator(splitting='V0 Vx R O R V1')
    al = ts.AlanineDipeptideExplicit()
    assert list(system_to_desc(al.system)['force_groups']) == [0] * 6
    assert list(system_to_desc(_grouped(al, bonded=1, nonbonded=0, reciprocal=2))['force_groups']) == [0, 1, 1, 1, 0, 2]


def test_oracle_equal_group_counts_reduce_to_the_single_time_step_scheme():
    """'V0 V1 R O R V1 V0' with every force in group 0 or 1: dt/2 f0 + dt/2 f1 = dt/2 f, the BAOAB trajectory."""
    al = ts.AlanineDipeptideVacuum() if hasattr(ts, 'AlanineDipeptideVacuum') else ts.AlanineDipeptideExplicit()
    out = []
    for system, splitting in ((al.system, 'V R O R V'), (_grouped(al), 'V0 V1 R O R V1 V0'), (_grouped(al), 'V1 V0 R O R V0 V1')):
        ora = OracleEngine(ForceFieldOracle)
        _setup(ora, system, al.positions, splitting, n_steps=3, dt=0.001)
        ora.propagate(2)
        out.append((ora.x.copy(), ora.v.copy()))
    for x, v in out[1:]:
        assert np.abs(x - out[0][0]).max() < 1e-10 and np.abs(v - out[0][1]).max() < 1e-8


@pytest.mark.gpu
def test_equal_group_counts_reduce_to_the_single_time_step_scheme_on_the_device(hip_engine_factory):
    al = ts.AlanineDipeptideExplicit()
    out = []
    for system, splitting in ((al.system, 'V R R O R R V'), (_grouped(al), 'V0 V1 R R O R R V1 V0'),
                              (_grouped(al, bonded=2, nonbonded=0, reciprocal=1), 'V0 V1 V2 R R O R R V2 V1 V0')):
        eng = hip_engine_factory()
        _setup(eng, system, al.positions, splitting, n_steps=8)
        assert not eng.propagate(4).any()
        out.append(eng.get_replicas()[:2])
    for x, v in out[1:]:
        # the groups' forces reach the velocities as separate fp32 kicks (and separate constraint projections)
        assert np.abs(x - out[0][0]).max() < 2e-5 and np.sqrt(((v - out[0][1]) ** 2).mean()) < 1e-3 * np.sqrt((out[0][1] ** 2).mean())


@pytest.mark.gpu
@pytest.mark.parametrize('groups,splitting', [
    (dict(bonded=1, nonbonded=0), SOLVENT_SOLUTE),                                     # the reference's docstring example
    (dict(bonded=0, nonbonded=1, reciprocal=2), 'V2 V1 V0 R V0 R O R V0 R V0 V1 V2'),   # mesh slowest, bonded fastest
])
def test_device_follows_the_oracle(hip_engine_factory, groups, splitting):
    """Alanine dipeptide in water, 6 steps at 2 fs: device fp32 vs oracle f64 on the same Philox stream, forces per group from
    the class mask of remd_compute_forces (direct space / mesh / listed terms separately)."""
    al = ts.AlanineDipeptideExplicit()
    system = _grouped(al, **groups)
    eng, ora = hip_engine_factory(), OracleEngine(ForceFieldOracle)
    for e in (eng, ora):
        _setup(e, system, al.positions, splitting, n_steps=6)
    assert not eng.propagate(3).any()
    ora.propagate(3)
    xg, vg = eng.get_replicas()[:2]
    assert np.abs(xg - ora.x).max() < 5e-5                      # nm (the single-time-step test allows the same after 10 steps)
    assert np.sqrt(((vg - ora.v) ** 2).mean()) < 2e-3 * np.sqrt((ora.v ** 2).mean())
    cons = ora.sys.constraints
    i, j, dist = np.array([c[0] for c in cons]), np.array([c[1] for c in cons]), np.array([c[2] for c in cons])
    assert np.abs(np.linalg.norm(xg[0][j] - xg[0][i], axis=1) - dist).max() < 3e-6


@pytest.mark.gpu
def test_a_force_class_in_an_unnamed_group_is_refused(hip_engine_factory):
    al = ts.AlanineDipeptideExplicit()
    eng = hip_engine_factory()
    _setup(eng, _grouped(al, bonded=1, nonbonded=0, reciprocal=3), al.positions, SOLVENT_SOLUTE, n_steps=2)
    with pytest.raises(RuntimeError, match='no V of the splitting names'):
        eng.propagate(0)


@pytest.mark.parametrize('groups,splitting', [
    (dict(bonded=1, nonbonded=0), SOLVENT_SOLUTE),                                     # the reference's docstring example
    (dict(bonded=0, nonbonded=1, reciprocal=2), 'V2 V1 V0 R V0 R O R V0 R V0 V1 V2'),   # mesh slowest, bonded fastest
])
def test_cpu_library_follows_the_oracle(groups, splitting):
    """The C++ port (libremd_cpu.so through the C ABI) on the same Philox stream as the f64 oracle: alanine dipeptide in water,
    3 steps at 2 fs, forces per group from the class mask (listed terms / direct space + exceptions / mesh separately); a V without a
    group in such a splitting and groups above 3 are refused as on the device.  (The four-atom chain of
    tests/test_integrator_program.py holds both to the reference's own step program.)"""
    import os
    import oracle
    from openmmtools_amd._engine import HipEngine
    lib = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')
    if not os.path.exists(lib):
        oracle.build()
    al = ts.AlanineDipeptideExplicit()
    system = _grouped(al, **groups)
    eng, ora = HipEngine(lib_path=lib), OracleEngine(ForceFieldOracle)
    for e in (eng, ora):
        _setup(e, system, al.positions, splitting, n_steps=3)
    assert not eng.propagate(2).any()
    ora.propagate(2)
    xc, vc = eng.get_replicas()[:2]
    assert np.abs(xc - ora.x).max() < 1e-7                      # nm: two f64 implementations with different constraint solvers / FFTs
    assert np.sqrt(((vc - ora.v) ** 2).mean()) < 1e-6 * np.sqrt((ora.v ** 2).mean())
    with pytest.raises(RuntimeError, match='must name the force group of every V'):
        eng.set_integrator('V0 V R O R V1', 0.002, 1.0, 2, True, 1e-8)
    with pytest.raises(RuntimeError, match='above 3'):
        eng.set_integrator('V0 V7 R O R V7 V0', 0.002, 1.0, 2, True, 1e-8)
    eng.set_integrator('V0 R O R V0', 0.002, 1.0, 2, True, 1e-8)                       # one group: plain V
    eng.close()


def test_host_parser_agrees_with_the_reference_parser_run_on_the_same_strings():
    """tests/golden/splittings_reference.json: verdicts of the reference's own _sanity_check / _parse_splitting_string
    (integrators.py:1319-1402, 1474-1537; executed by tests/golden/make_golden_splittings.py).  Same counts, same force-group
    table, same exception types; stricter on four strings the reference lets through by accident."""
    import json, os
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'splittings_reference.json')))['cases']
    stricter = {'O { V { R } V } O', 'O { V R O R V }', 'OR V', 'V R O 12'}       # nested braces, O inside them, multi-letter step names
    assert len(cases) == 31
    for c in cases:
        s = c['splitting']
        if c['ok'] and s not in stricter:
            integ = integrators.LangevinIntegrator(splitting=s)
            assert {k: integ._ORV_counts[k] for k in c['counts']} == c['counts'], s
            assert integ._mts == c['mts'] and integ._force_group_nV == c['n_v'], s
        else:
            with pytest.raises((ValueError, AssertionError)) as err:
                integrators.LangevinIntegrator(splitting=s)
            if not c['ok'] and c['error'] in ('ValueError', 'AssertionError'):
                assert type(err.value).__name__ == c['error'], (s, c['error'], err.value)
                if c['error'] == 'ValueError':
                    assert str(err.value) == c['message'], (s, str(err.value), c['message'])          # the sentence too
