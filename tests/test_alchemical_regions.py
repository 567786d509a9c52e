"""General alchemical regions: several named regions with their own lambdas, regions that interact
through the product of their lambdas, the 'direct-space' / 'coulomb' PME treatments and the reaction-field treatments with soft-core
electrostatics -- /root/reference/openmmtools/alchemy/alchemy.py:1356-1537 (expressions), :1539-2038 (force split).

Chain of evidence:
  * oracle/alchemical_regions.py (f64) reproduces the VALUES of the reference's own expression strings
    (tests/golden/reference_alchemy_expressions.json, produced by tests/golden/make_golden_alchemy_strings.py from the reference's
    syntax tree) pair by pair, through the factory of this package (openmmtools_amd.alchemy) -- so the factory's constants (alpha of the
    Ewald direct space, k_rf / c_rf, switching distances, mixing rules) are pinned too;
  * the C++ port (libremd_cpu.so, same C ABI) and -- under -m gpu -- the HIP kernels (csrc/alch_regions.hip) against that oracle on
    solvated systems: u_kl rows over a ladder of per-region lambdas, the own-state potential, the forces.
"""
import json
import os

import numpy as np
import pytest

import oracle
from oracle.alchemical_regions import RegionOracle, total_state_energies, total_energy_forces
from openmmtools_amd import alchemy, states, testsystems as ts
from openmmtools_amd.system import System, NonbondedForce, system_to_desc
from openmmtools_amd._engine import HipEngine

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, 'golden', 'reference_alchemy_expressions.json')))
KB = 0.008314462618153242
CPU_LIB = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')
L = 4.0


def _switch(r, rs, rc):
    """OpenMM's switching function of a CustomNonbondedForce (what multiplies the expression between rs and rc)"""
    if rs is None or r <= rs:
        return 1.0
    x = (r - rs) / (rc - rs)
    return 1.0 - 10.0 * x ** 3 + 15.0 * x ** 4 - 6.0 * x ** 5


def _pair_system(method, q1, q2, s1, s2, e1=0.0, e2=0.0):
    s = System()
    s.addParticle(12.0); s.addParticle(12.0)
    s.setDefaultPeriodicBoxVectors([L, 0, 0], [0, L, 0], [0, 0, L])
    nb = NonbondedForce()
    nb.setNonbondedMethod(method); nb.setCutoffDistance(1.0); nb.setEwaldErrorTolerance(5e-4); nb.setReactionFieldDielectric(78.3)
    nb.setUseSwitchingFunction(False); nb.setUseDispersionCorrection(False)
    nb.addParticle(q1, s1, e1); nb.addParticle(q2, s2, e2)
    s.addForce(nb)
    return s


def _region_value(system, r, ls, le):
    d = system_to_desc(system)
    reg = RegionOracle(d['alch_regions'], d['cutoff'], None, np.zeros((0, 2), int))
    x = np.array([[1.0, 1.3, 0.9], [1.0 + 0.6 * r, 1.3 - 0.48 * r, 0.9 + 0.64 * r]])
    return reg.energy_forces(x, [L, L, L], [ls], [le], forces=False)[0]


@pytest.mark.parametrize('key,method,kw,switched', [
    ('electrostatics_pme_direct_space', NonbondedForce.PME, dict(alchemical_pme_treatment='direct-space'), False),
    ('electrostatics_pme_coulomb', NonbondedForce.PME, dict(alchemical_pme_treatment='coulomb'), True),
    ('electrostatics_rf_switched', NonbondedForce.CutoffPeriodic, dict(alchemical_rf_treatment='switched'), True),
    ('electrostatics_rf_shifted', NonbondedForce.CutoffPeriodic, dict(alchemical_rf_treatment='shifted'), False),
])
def test_region_oracle_reproduces_the_reference_electrostatics_expressions(key, method, kw, switched):
    """the factory's electrostatic CustomNonbondedForce of an (environment, region) pair = the reference's expression string times
    OpenMM's switch (from cutoff - switch_width where the factory switches it on, alchemy.py:1818-1824)"""
    n = 0
    for smp in G['samples'][key]:
        if smp['r'] >= 1.0:
            continue
        region = alchemy.AlchemicalRegion(alchemical_atoms=[0], softcore_beta=smp['softcore_beta'], name='zero')
        system = alchemy.AbsoluteAlchemicalFactory(**kw).create_alchemical_system(
            _pair_system(method, smp['charge1'], smp['charge2'], smp['sigma1'], smp['sigma2']), region)
        assert system.alchemical_regions is not None and system.alchemical_region is None
        got = _region_value(system, smp['r'], 1.0, smp['lambda_electrostatics'])
        want = smp['value'] * _switch(smp['r'], 0.9 if switched else None, 1.0)
        assert np.isclose(got, want, rtol=1e-12, atol=1e-13), (key, smp, got, want)
        n += 1
    assert n >= 20


def test_region_oracle_reproduces_the_reference_sterics_expression_with_any_exponents():
    """sterics_random: random sigma / epsilon / lambda / softcore alpha, a, b (c = 6 there); through the general path because the
    alchemical atom is charged under 'direct-space' (charges of both atoms set so that the electrostatic part is known: none on the partner)"""
    n = 0
    for smp in G['samples']['sterics_random']:
        if smp['r'] >= 1.0:
            continue
        sc = smp['softcore']
        region = alchemy.AlchemicalRegion(alchemical_atoms=[0], softcore_alpha=sc['softcore_alpha'], softcore_a=sc['softcore_a'],
                                          softcore_b=sc['softcore_b'], softcore_c=sc['softcore_c'], name='zero')
        system = alchemy.AbsoluteAlchemicalFactory(alchemical_pme_treatment='direct-space').create_alchemical_system(
            _pair_system(NonbondedForce.PME, 0.3, 0.0, smp['sigma1'], smp['sigma2'], smp['epsilon1'], smp['epsilon2']), region)
        got = _region_value(system, smp['r'], smp['lambda_sterics'], 1.0)
        assert np.isclose(got, smp['value'], rtol=1e-12, atol=1e-13), (smp, got)
        n += 1
    assert n >= 50


def test_two_interacting_regions_use_the_product_of_their_lambdas():
    """alchemy.py:1368-1377 ('lambda_sterics_zero*lambda_sterics_one'): a pair of atoms of two interacting regions at (l0, l1) has the
    energy of an (environment, region) pair at l0 * l1; without the interaction the two regions do not see each other at all."""
    base = _pair_system(NonbondedForce.PME, 0.4, -0.3, 0.3, 0.34, 0.5, 0.7)
    regions = [alchemy.AlchemicalRegion(alchemical_atoms=[0], name='zero'), alchemy.AlchemicalRegion(alchemical_atoms=[1], name='one')]
    fac = alchemy.AbsoluteAlchemicalFactory(alchemical_pme_treatment='direct-space')
    both = fac.create_alchemical_system(base, regions, alchemical_regions_interactions=frozenset({(0, 1)}))
    apart = fac.create_alchemical_system(base, regions)
    single = fac.create_alchemical_system(base, alchemy.AlchemicalRegion(alchemical_atoms=[1], name='one'))
    d = system_to_desc(both)
    reg = RegionOracle(d['alch_regions'], 1.0, None, np.zeros((0, 2), int))
    x = np.array([[1.0, 1.3, 0.9], [1.3, 1.1, 1.2]])
    r = np.linalg.norm(x[1] - x[0])
    e = reg.energy_forces(x, [L, L, L], [0.6, 0.5], [0.8, 0.25], forces=False)[0]
    assert np.isclose(e, _region_value(single, r, 0.3, 0.2), rtol=1e-13)
    d0 = system_to_desc(apart)
    assert RegionOracle(d0['alch_regions'], 1.0, None, np.zeros((0, 2), int)).energy_forces(x, [L, L, L], [0.6, 0.5], [0.8, 0.25], forces=False)[0] == 0.0


def test_the_factory_chooses_the_path_and_refuses_what_the_reference_refuses():
    lj = ts.LennardJonesFluid(nparticles=64)
    fast = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, alchemy.AlchemicalRegion(alchemical_atoms=range(4)))
    assert fast.alchemical_region is not None and fast.alchemical_regions is None          # uncharged, c = 6: the pair kernels' own path
    general = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, alchemy.AlchemicalRegion(alchemical_atoms=range(4), softcore_c=12))
    assert general.alchemical_regions is not None and system_to_desc(general)['alch_regions']['electrostatics'] == 0
    al = ts.AlanineDipeptideExplicit()
    with pytest.raises(ValueError, match='Softcore electrostatics is not supported with exact treatment'):      # alchemy.py:1617-1625
        alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(22), softcore_beta=0.5))
    with pytest.raises(ValueError, match='Decoupled electrostatics is not supported with exact treatment'):
        alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(22), annihilate_electrostatics=False))
    with pytest.raises(ValueError, match='straddles two alchemical regions'):                                   # alchemy.py:1969
        alchemy.AbsoluteAlchemicalFactory(alchemical_pme_treatment='coulomb').create_alchemical_system(
            al.system, [alchemy.AlchemicalRegion(alchemical_atoms=range(6), name='a'), alchemy.AlchemicalRegion(alchemical_atoms=range(6, 22), name='b')])
    with pytest.raises(NotImplementedError, match='several charged alchemical regions'):
        alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(
            al.system, [alchemy.AlchemicalRegion(alchemical_atoms=range(22), name='a'), alchemy.AlchemicalRegion(alchemical_atoms=range(22, 25), name='b')])
    # the factory's NonbondedForce: alchemical atoms without charge and epsilon, their exceptions zeroed but kept (alchemy.py:1903-1911, 2001-2006)
    s = alchemy.AbsoluteAlchemicalFactory(alchemical_pme_treatment='coulomb').create_alchemical_system(al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(22), name='a'))
    nb = [f for f in s.getForces() if isinstance(f, NonbondedForce)][0]
    nb0 = [f for f in al.system.getForces() if isinstance(f, NonbondedForce)][0]
    assert all(nb.particles[i][0] == 0.0 and nb.particles[i][2] == 0.0 for i in range(22)) and nb.particles[22] == nb0.particles[22]
    assert len(nb.exceptions) == len(nb0.exceptions) and all(e[2] == 0.0 and e[4] == 0.0 for e in nb.exceptions if e[0] < 22 or e[1] < 22)
    assert nb0.particles[0][0] != 0.0                      # (the reference System is left alone)


# ---- engines against the oracle ---------------------------------------------------------------------------------------------
def _alanine_two_regions(kw, interactions, **region_kw):
    """the dipeptide ('pep') and three waters next to it ('wat') as two regions"""
    al = ts.AlanineDipeptideExplicit()
    regions = [alchemy.AlchemicalRegion(alchemical_atoms=range(22), name='pep', **region_kw),
               alchemy.AlchemicalRegion(alchemical_atoms=range(22, 31), name='wat', annihilate_sterics=True, softcore_alpha=0.4)]
    system = alchemy.AbsoluteAlchemicalFactory(**kw).create_alchemical_system(al.system, regions, alchemical_regions_interactions=interactions)
    return al, system, regions


LADDER_S = np.array([[1.0, 1.0], [1.0, 0.6], [0.7, 1.0], [0.35, 0.8], [0.0, 0.5], [0.0, 0.0]])
LADDER_E = np.array([[1.0, 1.0], [0.5, 1.0], [0.0, 0.7], [0.0, 0.3], [0.0, 0.0], [0.0, 0.0]])


def _check_engine_against_the_oracle(eng, kw, interactions, rtol, ftol, region_kw=None):
    al, system, regions = _alanine_two_regions(kw, interactions, **(region_kw or {}))
    nb = [f for f in system.getForces() if isinstance(f, NonbondedForce)][0]
    box0 = np.diag(system.getDefaultPeriodicBoxVectors())
    econst = alchemy.alchemical_long_range_constants(system, nb, LADDER_S, float(np.prod(box0)))
    assert np.all(np.isfinite(econst)) and econst[0] != econst[-1]
    desc = system_to_desc(system, ewald_split='reference')
    eng.set_system(desc)
    K = len(LADDER_S)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(K, beta), None, None, econst)
    eng.set_region_lambdas(LADDER_S, LADDER_E)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 5, True, 1e-8)
    eng.seed(7)
    labels = np.array([0, 3, 4])
    x = np.stack([al.positions + 0.001 * r * np.random.default_rng(r).normal(size=al.positions.shape) for r in range(3)])
    box = np.tile(box0, (3, 1))
    eng.set_replicas(3, 0, x, None, box, labels)
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    for r, k in enumerate(labels):
        ref = total_state_energies(desc, xd[r], box[r], LADDER_S, LADDER_E)
        assert np.ptp(ref) > 10.0                                     # the ladder matters: tens of kJ/mol between its ends
        assert np.allclose(rows[r], beta * (ref + econst), rtol=rtol), np.abs(rows[r] / (beta * (ref + econst)) - 1).max()
        assert np.isclose(U[r], ref[k], rtol=rtol)
        f_ref = total_energy_forces(desc, xd[r], box[r], LADDER_S[k], LADDER_E[k])[1]
        assert np.abs(f[r] - f_ref).max() < ftol * np.abs(f_ref).max(), np.abs(f[r] - f_ref).max() / np.abs(f_ref).max()
    return eng


CASES = [
    (dict(alchemical_pme_treatment='direct-space'), frozenset({(0, 1)}), dict(softcore_beta=0.3)),
    (dict(alchemical_pme_treatment='coulomb', switch_width=0.15), frozenset(), dict(softcore_c=4, softcore_a=2, softcore_f=4, softcore_e=2, softcore_beta=0.2)),
]


@pytest.mark.parametrize('kw,interactions,region_kw', CASES)
def test_cpu_port_matches_the_region_oracle(kw, interactions, region_kw):
    if not os.path.exists(CPU_LIB):
        oracle.build()
    eng = _check_engine_against_the_oracle(HipEngine(lib_path=CPU_LIB), kw, interactions, 1e-9, 1e-8, region_kw)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('kw,interactions,region_kw', CASES)
def test_hip_regions_match_the_region_oracle(hip_engine_factory, kw, interactions, region_kw):
    """csrc/alch_regions.hip: u_kl rows over the ladder, the own-state potential (1e-5) and the forces against the f64 oracle"""
    eng = _check_engine_against_the_oracle(hip_engine_factory(), kw, interactions, 1e-5, 2e-4, region_kw)
    # ... and the MD loop runs on it (one block of replicas: phases are off with regions)
    nan = eng.propagate(0)
    assert not np.any(nan)
    rows2 = eng.compute_energies()
    assert np.all(np.isfinite(rows2))
