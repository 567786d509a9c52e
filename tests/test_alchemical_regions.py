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
from openmmtools_amd import alchemy, states, system_xml, _alchemical_xml as ax, testsystems as ts
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
    """The class of interacting regions of the C ABI (remd_alch_regions_desc.interactions; alchemy.py:1368-1377,
    'lambda_sterics_zero*lambda_sterics_one'): a pair of atoms of two interacting regions at (l0, l1) has the energy of an
    (environment, region) pair at l0 * l1; without the interaction the two regions do not see each other at all.

    The reference's factory, AS ITS LOOP EXECUTES, never reaches that class: it builds the forces of a pair of regions from particle
    tables it has zeroed in the two regions' single turns (alchemy.py:1693, 1886-1911), so this package's factory passes no interacting
    pairs to the engine either -- alchemical_regions_interactions changes nothing outside the exact PME treatment."""
    base = _pair_system(NonbondedForce.PME, 0.4, -0.3, 0.3, 0.34, 0.5, 0.7)
    regions = [alchemy.AlchemicalRegion(alchemical_atoms=[0], name='zero'), alchemy.AlchemicalRegion(alchemical_atoms=[1], name='one')]
    fac = alchemy.AbsoluteAlchemicalFactory(alchemical_pme_treatment='direct-space')
    both = fac.create_alchemical_system(base, regions, alchemical_regions_interactions=frozenset({(0, 1)}))
    apart = fac.create_alchemical_system(base, regions)
    single = fac.create_alchemical_system(base, alchemy.AlchemicalRegion(alchemical_atoms=[1], name='one'))
    x = np.array([[1.0, 1.3, 0.9], [1.3, 1.1, 1.2]])
    r = np.linalg.norm(x[1] - x[0])
    energy = lambda terms: RegionOracle(terms, 1.0, None, np.zeros((0, 2), int)).energy_forces(x, [L, L, L], [0.6, 0.5], [0.8, 0.25], forces=False)[0]
    assert both.alchemical_regions_interactions == [(0, 1)]
    assert energy(system_to_desc(both)['alch_regions']) == 0.0 and energy(system_to_desc(apart)['alch_regions']) == 0.0
    terms = dict(system_to_desc(apart)['alch_regions'], interactions=np.array([[1, 2]], dtype=np.int32))
    assert np.isclose(energy(terms), _region_value(single, r, 0.3, 0.2), rtol=1e-13)


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
    # an exception between two regions is a bond of the FIRST region's (environment, region) force (alchemy.py:1972-1976, 1992-2006)
    cut = alchemy.AbsoluteAlchemicalFactory(alchemical_pme_treatment='coulomb').create_alchemical_system(
        al.system, [alchemy.AlchemicalRegion(alchemical_atoms=range(6), name='a'), alchemy.AlchemicalRegion(alchemical_atoms=range(6, 22), name='b')])
    d = system_to_desc(cut)
    reg = RegionOracle(d['alch_regions'], d['cutoff'], d['switch_distance'], d['exception_atoms'])
    straddling = [k for k, (i, j) in enumerate(reg.exc_atoms) if (i < 6) != (j < 6)]
    assert len(straddling) > 5 and all(reg.exc_kinds[k] == (0, 1, 1, 1) for k in straddling)
    # the default reaction-field treatment re-writes the WHOLE system's reaction field (alchemy.py:744-749): carried to the engine
    rfs = alchemy.AbsoluteAlchemicalFactory(switch_width=0.12).create_alchemical_system(_charged_lj_fluid(), alchemy.AlchemicalRegion(alchemical_atoms=range(4)))
    assert rfs.alchemical_regions is not None and system_to_desc(rfs)['rf_unshifted_switch_width'] == 0.12
    assert 'rf_unshifted_switch_width' not in system_to_desc(alchemy.AbsoluteAlchemicalFactory(alchemical_rf_treatment='shifted').create_alchemical_system(
        _charged_lj_fluid(), alchemy.AlchemicalRegion(alchemical_atoms=range(4))))
    # several charged regions under the exact PME treatment (the default): the regions' charges as parameter offsets (alchemy.py:1675-1680)
    ex = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(
        al.system, [alchemy.AlchemicalRegion(alchemical_atoms=range(22), name='a'), alchemy.AlchemicalRegion(alchemical_atoms=range(22, 25), name='b')],
        alchemical_regions_interactions=frozenset({(0, 1)}))
    t = system_to_desc(ex)['alch_regions']
    assert t['exact_pme'] == 1 and t['electrostatics'] == 0 and t['interactions'].tolist() == [[1, 2]] and t['charge'][0] != 0.0
    assert system_to_desc(ex)['charge'][0] == 0.0
    # the factory's NonbondedForce: alchemical atoms without charge and epsilon, their exceptions zeroed but kept (alchemy.py:1903-1911, 2001-2006)
    s = alchemy.AbsoluteAlchemicalFactory(alchemical_pme_treatment='coulomb').create_alchemical_system(al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(22), name='a'))
    nb = [f for f in s.getForces() if isinstance(f, NonbondedForce)][0]
    nb0 = [f for f in al.system.getForces() if isinstance(f, NonbondedForce)][0]
    assert all(nb.particles[i][0] == 0.0 and nb.particles[i][2] == 0.0 for i in range(22)) and nb.particles[22] == nb0.particles[22]
    assert len(nb.exceptions) == len(nb0.exceptions) and all(e[2] == 0.0 and e[4] == 0.0 for e in nb.exceptions if e[0] < 22 or e[1] < 22)
    assert nb0.particles[0][0] != 0.0                      # (the reference System is left alone)


def _charged_lj_fluid(n=216):
    """a Lennard-Jones fluid with alternating charges under CutoffPeriodic (reaction field)"""
    import copy
    lj = ts.LennardJonesFluid(nparticles=n, reduced_density=0.4)
    system = copy.deepcopy(lj.system)
    nb = [f for f in system.getForces() if isinstance(f, NonbondedForce)][0]
    for i, (q, s_, e) in enumerate(nb.particles):
        nb.particles[i] = (0.25 if i % 2 == 0 else -0.25, s_, e)
    return system


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


def _check_engine_against_the_oracle(eng, kw, interactions, rtol, ftol, region_kw=None, ewald_split='reference', labels=(0, 3, 4), check=True):
    """interactions: pairs of regions handed to the ENGINE as interacting (the factory itself passes none on, see above)"""
    al, system, regions = _alanine_two_regions(kw, interactions, **(region_kw or {}))
    nb = [f for f in system.getForces() if isinstance(f, NonbondedForce)][0]
    box0 = np.diag(system.getDefaultPeriodicBoxVectors())
    econst = alchemy.alchemical_long_range_constants(system, nb, LADDER_S, float(np.prod(box0)))
    assert np.all(np.isfinite(econst)) and econst[0] != econst[-1]
    desc = system_to_desc(system, ewald_split=ewald_split)
    if interactions:
        desc['alch_regions']['interactions'] = np.array([(a + 1, b + 1) for a, b in sorted(interactions)], dtype=np.int32)
    eng.set_system(desc)
    K = len(LADDER_S)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(K, beta), None, None, econst)
    eng.set_region_lambdas(LADDER_S, LADDER_E)
    terms = desc['alch_regions']
    BONDED = None
    if any(len(terms.get(k + '_atoms', ())) for k in ('bond', 'angle', 'torsion')):
        # softened bonded terms of the dipeptide (alchemy.py:1115-1354): their own ladders
        assert len(terms['torsion_atoms']) > 10 and len(terms['angle_atoms']) > 10 and len(terms['bond_atoms']) == 3
        assert len(system_to_desc(al.system)['torsion_atoms']) == len(desc['torsion_atoms']) + len(terms['torsion_atoms'])
        BONDED = np.ones((K, 3, 2))
        BONDED[:, 0, 0] = [1.0, 0.9, 0.7, 0.5, 0.2, 0.0]; BONDED[:, 1, 0] = [1.0, 1.0, 0.8, 0.6, 0.3, 0.1]; BONDED[:, 2, 0] = [1.0, 0.5, 0.5, 0.25, 0.0, 0.0]
        eng.set_region_bonded_lambdas(BONDED[:, 0], BONDED[:, 1], BONDED[:, 2])
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 5, True, 1e-8)
    eng.seed(7)
    labels = np.array(labels)
    nr = len(labels)
    x = np.stack([al.positions + 0.001 * r * np.random.default_rng(r).normal(size=al.positions.shape) for r in range(nr)])
    box = np.tile(box0, (nr, 1))
    eng.set_replicas(nr, 0, x, None, box, labels)
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    for r, k in enumerate(labels if check else ()):
        ref = total_state_energies(desc, xd[r], box[r], LADDER_S, LADDER_E, BONDED)
        assert np.ptp(ref) > 10.0                                     # the ladder matters: tens of kJ/mol between its ends
        assert np.allclose(rows[r], beta * (ref + econst), rtol=rtol), np.abs(rows[r] / (beta * (ref + econst)) - 1).max()
        assert np.isclose(U[r], ref[k], rtol=rtol)
        f_ref = total_energy_forces(desc, xd[r], box[r], LADDER_S[k], LADDER_E[k], (None, None, None) if BONDED is None else tuple(BONDED[k]))[1]
        assert np.abs(f[r] - f_ref).max() < ftol * np.abs(f_ref).max(), np.abs(f[r] - f_ref).max() / np.abs(f_ref).max()
    return eng


CASES = [
    (dict(alchemical_pme_treatment='direct-space'), frozenset({(0, 1)}), dict(softcore_beta=0.3)),
    (dict(alchemical_pme_treatment='coulomb', switch_width=0.15), frozenset(), dict(softcore_c=4, softcore_a=2, softcore_f=4, softcore_e=2, softcore_beta=0.2)),
    # the exact PME treatment (the factory's default): every region's charges scaled by its own lambda inside the whole Ewald sum,
    # regions that do not interact excluded from each other / regions that do see each other's scaled charges (alchemy.py:1663-1681)
    (dict(), frozenset(), dict(softcore_c=8)),
    (dict(), frozenset({(0, 1)}), dict()),
    # consistent_exceptions=True (alchemy.py:1456-1461): the exceptions' electrostatics with the pairs' erfc(alpha r_eff) / r_eff
    (dict(alchemical_pme_treatment='direct-space', consistent_exceptions=True), frozenset(), dict(softcore_beta=0.25)),
    # softened bonded terms: every angle and proper torsion of the dipeptide and three of its bonds under lambda_angles_pep / lambda_torsions_pep /
    # lambda_bonds_pep (alchemy.py:1115-1354)
    (dict(), frozenset(), dict(alchemical_torsions=True, alchemical_angles=True, alchemical_bonds=[0, 2, 5])),
]


@pytest.mark.parametrize('kw,interactions,region_kw', CASES)
def test_cpu_port_matches_the_region_oracle(kw, interactions, region_kw):
    if not os.path.exists(CPU_LIB):
        oracle.build()
    # (two replicas here: the f64 port evaluates the whole Ewald sum once per state under the exact treatment; three on the device)
    eng = _check_engine_against_the_oracle(HipEngine(lib_path=CPU_LIB), kw, interactions, 1e-9, 1e-8, region_kw, labels=(3, 4))
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('kw,interactions,region_kw', CASES)
def test_hip_regions_match_the_region_oracle(hip_engine_factory, kw, interactions, region_kw):
    """csrc/alch_regions.hip: u_kl rows over the ladder, the own-state potential (1e-5) and the forces against the f64 oracle"""
    # (the device's own split of the Ewald sum: a Coulomb range beyond the cutoff -- the custom forces keep the NonbondedForce's cutoff, the
    # exact treatment's direct-space terms follow the split)
    eng = _check_engine_against_the_oracle(hip_engine_factory(), kw, interactions, 1e-5, 2e-4, region_kw, ewald_split='auto')
    # ... and the MD loop runs on it (one block of replicas: phases are off with regions)
    nan = eng.propagate(0)
    assert not np.any(nan)
    rows2 = eng.compute_energies()
    assert np.all(np.isfinite(rows2))


# ---- the store adapter: the reference's force set written and read back ------------------------------------------------------------
def _same_description(a, b):
    da, db = system_to_desc(a), system_to_desc(b)
    assert sorted(da) == sorted(db)
    for k in da:
        if k == 'alch_regions':
            assert sorted(da[k]) == sorted(db[k])
            for q in da[k]:
                if not q.endswith('_index'):                   # (where a softened term sat in the reference's force)
                    assert np.array_equal(np.asarray(da[k][q]), np.asarray(db[k][q])), q
        else:
            assert np.array_equal(np.asarray(da[k]), np.asarray(db[k])), k


@pytest.mark.gpu
@pytest.mark.parametrize('kw,interactions,region_kw', [CASES[0], CASES[2], CASES[5]])
def test_regions_in_two_phases_are_the_regions_in_one_block(hip_engine_factory, kw, interactions, region_kw):
    """A handle with general alchemical regions runs its replicas as two blocks like any other (round 6: the blocks get the regions'
    descriptor, the states' lambdas and the bonded lambdas from the handle: remd_regions_clone): eight replicas over the ladder --
    'direct-space', the exact PME treatment, softened bonded terms -- positions, velocities and u_kl after two propagations equal the
    one-block run bit for bit."""
    out = []
    for phases in (1, 2):
        eng = hip_engine_factory()
        eng.set_phases(phases)
        _check_engine_against_the_oracle(eng, kw, interactions, 1e-5, 2e-4, region_kw, ewald_split='auto', labels=(0, 4, 5, 0, 4, 5, 4, 0), check=False)        # (the ladder's middle states soften the sterics with the charges on: not for dynamics)
        eng.set_integrator('V R R O R R V', 0.002, 1.0, 40, True, 1e-8)
        for it in range(2):
            assert not eng.propagate(it).any()
            u = eng.compute_energies()
        x, v = eng.get_replicas()[:2]
        out.append((x.copy(), v.copy(), u.copy(), eng.phases_active()))
    (xa, va, ua, pa), (xb, vb, ub, pb) = out
    assert (pa, pb) == (1, 2)
    assert np.array_equal(xa, xb) and np.array_equal(va, vb) and np.array_equal(ua, ub)


def test_written_literals_of_the_other_treatments_are_the_references():
    """the expression strings this package writes for the 'direct-space' / 'coulomb' PME treatments and the shifted reaction field,
    character for character (alchemy.py:1392-1537 executed from the reference's syntax tree: tests/golden/make_golden_alchemy_strings.py)"""
    E = G['expressions']
    nb = NonbondedForce()
    nb.setNonbondedMethod(NonbondedForce.PME); nb.setCutoffDistance(1.0); nb.setEwaldErrorTolerance(5e-4)
    assert ax.electrostatics_expressions(nb, pme_treatment='direct-space') == (E['electrostatics_pme_direct_space'], E['electrostatics_exception_rf'])
    assert ax.electrostatics_expressions(nb, pme_treatment='coulomb')[0] == E['electrostatics_pme_coulomb']
    nb.setNonbondedMethod(NonbondedForce.CutoffPeriodic); nb.setReactionFieldDielectric(78.3)
    assert ax.electrostatics_expressions(nb, rf_treatment='shifted')[0] == E['electrostatics_rf_shifted']
    assert ax.electrostatics_expressions(nb, rf_treatment='switched')[0] == E['electrostatics_rf_switched']
    assert ax.electrostatics_expressions(nb, rf_treatment='switched', consistent_exceptions=True)[1] == E['electrostatics_exception_rf_consistent']


def test_written_literals_of_the_bonded_custom_forces_and_the_unshifted_reaction_field_are_the_references():
    """tests/golden/reference_bonded_expressions.json (make_golden_bonded_strings.py: the f-strings of alchemy.py:1115-1354 and the pieces of
    forces.UnshiftedReactionFieldForce's energy expression, forces.py:1128-1152, out of the reference's syntax tree)"""
    B = json.load(open(os.path.join(HERE, 'golden', 'reference_bonded_expressions.json')))
    for kind, (expr, per) in ax._BONDED_CUSTOM.items():
        ref = B['bonded'][kind]
        assert expr % 'LAMBDA' == ref['energy'] and list(per) == ref['per_term_parameters'] and ref['lambda_base_name'] == 'lambda_%ss' % kind
    pieces = B['unshifted_reaction_field']['energy_pieces']
    assert ax._RF_HEAD == pieces[0] and pieces[1] == 'chargeprod = charge1*charge2;' and B['unshifted_reaction_field']['per_particle_parameters'] == ['charge']
    # the document this package writes: the same pieces, k_rf and ONE_4PI_EPS0 formatted ':f' as the reference does
    lj = _charged_lj_fluid()
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj, alchemy.AlchemicalRegion(alchemical_atoms=range(4)))
    import xml.etree.ElementTree as ET
    rf = ET.fromstring(system_xml.to_xml(system)).find('Forces').findall('Force')[-1]
    nb = [f for f in lj.getForces() if isinstance(f, NonbondedForce)][0]
    k_rf = nb.getCutoffDistance() ** -3 * (78.3 - 1.0) / (2.0 * 78.3 + 1.0)
    assert rf.get('energy') == pieces[0] + pieces[1] + 'k_rf = %f;' % k_rf + 'ONE_4PI_EPS0 = %f;' % G['ONE_4PI_EPS0']
    assert [p.get('name') for p in rf.find('PerParticleParameters')] == ['charge'] and rf.get('useSwitchingFunction') == '1'


@pytest.mark.parametrize('kw,interactions,region_kw', CASES + [(dict(alchemical_pme_treatment='coulomb'), frozenset({(0, 1)}), {})])
def test_general_regions_are_written_as_the_factorys_force_set_and_read_back(kw, interactions, region_kw):
    al, system, regions = _alanine_two_regions(kw, interactions, **region_kw)
    xml = system_xml.to_xml(system)
    back, barostat = system_xml.from_xml(xml)
    assert barostat is None and [r.name for r in back.alchemical_regions] == ['pep', 'wat']
    assert back.alchemical_regions_interactions == sorted(interactions) and back.alchemical_factory_options == system.alchemical_factory_options
    softened = ('alchemical_bonds', 'alchemical_angles', 'alchemical_torsions')
    for r0, r1 in zip(system.alchemical_regions, back.alchemical_regions):
        # (softened bonded terms come back as indices into the re-assembled plain forces: the reference's own indices are not in the document)
        assert {k: v for k, v in r0.__dict__.items() if k not in softened} == {k: v for k, v in r1.__dict__.items() if k not in softened}
        assert all(bool(getattr(r0, k)) == bool(getattr(r1, k)) for k in softened)
    _same_description(system, back)
    # the document: per region 4 electrostatics forces (group of lambda_electrostatics_<name>) and 4 sterics forces, sorted by lambda
    # name (alchemy.py:1075-1083); a pair of interacting regions adds a nonbonded + a bond force to the FIRST region's lists (:2027-2032)
    import xml.etree.ElementTree as ET
    forces = ET.fromstring(xml).find('Forces').findall('Force')
    custom = [f for f in forces if f.get('type').startswith('Custom')]
    if region_kw.get('alchemical_torsions'):
        # softened bonded terms: one Custom{Angle,Bond,Torsion}Force of the region, energy lambda x the reference term, its lambda the only
        # global parameter; lambda names sorted: angles < bonds < electrostatics < sterics < torsions (alchemy.py:1075-1083, 1170-1197, 1252-1275, 1331-1354)
        assert [f.get('type') for f in custom[:2]] == ['CustomAngleForce', 'CustomBondForce'] and custom[-1].get('type') == 'CustomTorsionForce'
        assert custom[0].get('energy') == 'lambda_angles_pep*(K/2)*(theta-theta0)^2;' and custom[1].get('energy') == 'lambda_bonds_pep*(K/2)*(r-r0)^2;'
        assert custom[-1].get('energy') == 'lambda_torsions_pep*k*(1+cos(periodicity*theta-phase))'
        assert [g.get('name') for g in custom[-1].find('GlobalParameters')] == ['lambda_torsions_pep'] and len(custom[1].find('Bonds')) == 3
        groups = [int(f.get('forceGroup')) for f in custom]
        assert groups == sorted(groups)
        custom = custom[2:-1]
    extra = 2 if interactions else 0
    if kw.get('alchemical_pme_treatment', 'exact') == 'exact':
        # no electrostatic custom forces: per region a global parameter + particle / exception offsets of the NonbondedForce, which sits in the
        # group of the LAST lambda_electrostatics the loop touched (alchemy.py:1675-1681, 1893-1899, 2034-2035); regions that do not interact
        # exclude each other (:1663-1672)
        assert len(custom) == 8 + extra
        nbf = [f for f in forces if f.get('type') == 'NonbondedForce'][0]
        assert sorted(g.get('name') for g in nbf.find('GlobalParameters')) == ['lambda_electrostatics_pep', 'lambda_electrostatics_wat']
        offs = nbf.find('ParticleOffsets').findall('Offset')
        assert [int(o.get('particle')) for o in offs] == list(range(31)) and offs[0].get('parameter') == 'lambda_electrostatics_pep' and offs[30].get('parameter') == 'lambda_electrostatics_wat'
        n_exc = len(nbf.find('Exceptions'))
        n_ref = len([f for f in al.system.getForces() if isinstance(f, NonbondedForce)][0].exceptions)
        assert n_exc == n_ref + (0 if interactions else 22 * 9)
        # lambda names sorted, a force group each (:1075-1083): electrostatics_pep, electrostatics_wat, sterics_pep, sterics_wat; the loop's
        # last turn is the pair (pep, wat) when the regions interact, else wat
        assert int(nbf.get('forceGroup')) == int(custom[0].get('forceGroup')) - (2 if interactions else 1)
        return
    assert len(custom) == 16 + 2 * extra
    groups = [int(f.get('forceGroup')) for f in custom]
    assert groups == sorted(groups) and len(set(groups)) == 4
    names = [sorted(g.get('name') for g in f.find('GlobalParameters') if g.get('name').startswith('lambda')) for f in custom]
    assert names[0] == ['lambda_electrostatics_pep'] and names[4 + extra][0] == 'lambda_electrostatics_wat' and names[8 + extra][0] == 'lambda_sterics_pep'
    if interactions:
        pair = custom[4]                                    # (pep, wat) electrostatics: both lambdas, particle table zeroed in the single turns
        assert names[4] == ['lambda_electrostatics_pep', 'lambda_electrostatics_wat'] and 'lambda_electrostatics_pep*lambda_electrostatics_wat' in pair.get('energy')
        assert all(float(p.get('param1')) == 0.0 for k, p in enumerate(pair.find('Particles')) if k < 31)
    # 'wat' annihilates its sterics, 'pep' does not: lambda fixed in pep's alchemical/alchemical forces only
    aa_pep = custom[8 + extra + 1]
    assert aa_pep.get('energy').endswith('lambda_sterics_pep=1.0;') and names[8 + extra + 1] == []


def test_a_named_single_region_and_a_charged_region_under_the_shifted_reaction_field_round_trip():
    lj = _charged_lj_fluid()
    fac = alchemy.AbsoluteAlchemicalFactory(alchemical_rf_treatment='shifted')
    system = fac.create_alchemical_system(lj, alchemy.AlchemicalRegion(alchemical_atoms=range(6), name='ligand', softcore_beta=0.25, annihilate_electrostatics=False))
    assert system.alchemical_regions is not None
    back, _ = system_xml.from_xml(system_xml.to_xml(system))
    assert back.alchemical_regions[0].name == 'ligand' and not back.alchemical_regions[0].annihilate_electrostatics
    assert back.alchemical_factory_options['alchemical_rf_treatment'] == 'shifted'
    _same_description(system, back)
    # the states of such a System are reached by the region's name (alchemy.py:203-231)
    st = states.AlchemicalState.from_system(back, parameters_name_suffix='ligand')
    st.lambda_sterics_ligand = 0.5
    assert st.lambda_sterics == 0.5
    with pytest.raises(states.AlchemicalStateError):
        states.AlchemicalState.from_system(back, parameters_name_suffix='other')


# ---- the sampler on a System with two named regions (C++ port of the ABI here; the device under -m gpu) --------------------------------
from openmmtools_amd import mcmc, unit                                                        # noqa: E402
from openmmtools_amd.multistate import ReplicaExchangeSampler, MultiStateReporter, _hdf5      # noqa: E402

needs_hdf5 = pytest.mark.skipif(not _hdf5.available(), reason='libhdf5 not loadable')
LAM_A = [(1.0, 1.0), (1.0, 0.5), (0.6, 0.0), (0.0, 0.0)]          # (lambda_sterics_a, lambda_electrostatics_a)
LAM_B = [(1.0, 1.0), (0.8, 1.0), (0.8, 0.3), (0.2, 0.0)]


def _two_region_sampler(engine, storage, n_iterations, comm=None):
    lj = _charged_lj_fluid()
    plain = ts.LennardJonesFluid(nparticles=216, reduced_density=0.4)
    regions = [alchemy.AlchemicalRegion(alchemical_atoms=range(4), name='a', softcore_beta=0.2),
               alchemy.AlchemicalRegion(alchemical_atoms=range(4, 8), name='b', annihilate_sterics=True)]
    asys = alchemy.AbsoluteAlchemicalFactory(alchemical_rf_treatment='shifted').create_alchemical_system(lj, regions)
    ths = [states.CompoundThermodynamicState(states.ThermodynamicState(asys, 120.0 * unit.kelvin),
                                             [states.AlchemicalState(parameters_name_suffix='a', lambda_sterics_a=a[0], lambda_electrostatics_a=a[1]),
                                              states.AlchemicalState(parameters_name_suffix='b', lambda_sterics_b=b[0], lambda_electrostatics_b=b[1])])
           for a, b in zip(LAM_A, LAM_B)]
    ss = states.SamplerState(plain.positions, box_vectors=plain.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, n_steps=3, reassign_velocities=True, splitting='V R O R V')
    s = ReplicaExchangeSampler(mcmc_moves=move, number_of_iterations=n_iterations, engine=engine, seed=3, online_analysis_interval=None,
                               **({} if comm is None else dict(comm=comm)))
    s.create(ths, [ss], storage=storage)
    return s, asys


def _check_sampler_rows(s, asys, rtol):
    """u_kl of the sampler's first energy pass against the oracle: every state's own (lambda_a, lambda_b) reaches the engine"""
    s._compute_energies()
    desc = system_to_desc(asys)
    nb = [f for f in asys.getForces() if isinstance(f, NonbondedForce)][0]
    LS = np.array([[a[0], b[0]] for a, b in zip(LAM_A, LAM_B)]); LE = np.array([[a[1], b[1]] for a, b in zip(LAM_A, LAM_B)])
    box = np.diag(asys.getDefaultPeriodicBoxVectors())
    econst = alchemy.alchemical_long_range_constants(asys, nb, LS, float(np.prod(box)))
    x = s._engine.get_replicas()[0]
    for r in range(len(LAM_A)):
        ref = (total_state_energies(desc, x[r], box, LS, LE) + econst) / (KB * 120.0)
        assert np.allclose(s.energy_thermodynamic_states[r], ref, rtol=rtol, atol=1e-6 * np.abs(ref).max()), np.abs(s.energy_thermodynamic_states[r] - ref).max()


@needs_hdf5
def test_sampler_with_two_named_regions_stores_in_the_references_layout_and_resumes(tmp_path):
    if not os.path.exists(CPU_LIB):
        oracle.build()
    rep = MultiStateReporter(str(tmp_path / 'two.nc'), checkpoint_interval=1)
    s, asys = _two_region_sampler(HipEngine(lib_path=CPU_LIB), rep, 2)
    _check_sampler_rows(s, asys, 1e-9)
    s.run()
    rep.close()
    r = MultiStateReporter(str(tmp_path / 'two.nc'), open_mode='r')
    d = r.read_dict('thermodynamic_states/state2')
    assert [c['parameters_name_suffix'] for c in d['composable_states']] == ['a', 'b']                    # states.py:3879-3898
    assert d['composable_states'][0]['parameters']['lambda_sterics'] == 0.6 and d['composable_states'][1]['parameters']['lambda_electrostatics'] == 0.3
    th, _ = r.read_thermodynamic_states()
    assert [(t.lambda_sterics_a, t.lambda_electrostatics_a) for t in th] == LAM_A and [(t.lambda_sterics_b, t.lambda_electrostatics_b) for t in th] == LAM_B
    assert th[0].system.fingerprint() == asys.fingerprint()
    r.close()
    full, _ = _two_region_sampler(HipEngine(lib_path=CPU_LIB), MultiStateReporter(str(tmp_path / 'full.nc'), checkpoint_interval=1), 4)
    full.run()
    full._reporter.close()
    res = ReplicaExchangeSampler.from_storage(str(tmp_path / 'two.nc'), engine=HipEngine(lib_path=CPU_LIB))
    assert res.iteration == 2
    res.extend(2)
    res._reporter.close()
    ea = MultiStateReporter(str(tmp_path / 'two.nc'), open_mode='r').read_energies()[0]
    eb = MultiStateReporter(str(tmp_path / 'full.nc'), open_mode='r').read_energies()[0]
    assert ea.shape == eb.shape == (5, 4, 4) and np.array_equal(ea[:3], eb[:3]) and np.allclose(ea[3:], eb[3:], rtol=2e-5, atol=1e-5)
    assert np.ptp(ea[0][0]) > 1.0                                              # the ladder matters


@pytest.mark.gpu
def test_sampler_with_two_named_regions_on_the_device(hip_engine_factory):
    s, asys = _two_region_sampler(hip_engine_factory(), None, 3)
    _check_sampler_rows(s, asys, 2e-5)
    s.run()
    assert np.all(np.isfinite(s.energy_thermodynamic_states)) and s.iteration == 3


# ---- the reference's own multi-region case (tests/test_alchemy.py:2203-2208, 2216-2252): CB7:B2 with the guest as region 'zero' and
# atoms 156-159 (a water and the next water's oxygen) as region 'one', under the exact PME treatment and under 'direct-space' -----------
def _host_guest_two_regions(kw):
    hg = ts.HostGuestExplicit()
    regions = [alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156), name='zero'), alchemy.AlchemicalRegion(alchemical_atoms=range(156, 160), name='one')]
    return hg, alchemy.AbsoluteAlchemicalFactory(**kw).create_alchemical_system(hg.system, regions)


HG_S = np.array([[1.0, 1.0], [1.0, 1.0], [0.6, 1.0], [0.0, 0.4]])
HG_E = np.array([[1.0, 1.0], [0.5, 0.0], [0.0, 0.0], [0.0, 0.0]])


def _check_host_guest(eng, kw, rtol, ftol, ewald_split):
    hg, system = _host_guest_two_regions(kw)
    nb = [f for f in system.getForces() if isinstance(f, NonbondedForce)][0]
    box0 = np.diag(system.getDefaultPeriodicBoxVectors())
    econst = alchemy.alchemical_long_range_constants(system, nb, HG_S, float(np.prod(box0)))
    desc = system_to_desc(system, ewald_split=ewald_split)
    eng.set_system(desc)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(4, beta), None, None, econst)
    eng.set_region_lambdas(HG_S, HG_E)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 3, True, 1e-8)
    eng.seed(11)
    labels = np.array([1, 3])
    x = np.stack([hg.positions + 0.001 * (r + 1) * np.random.default_rng(r).normal(size=hg.positions.shape) for r in range(2)])
    box = np.tile(box0, (2, 1))
    eng.set_replicas(2, 0, x, None, box, labels)
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    for r, k in enumerate(labels):
        ref = total_state_energies(desc, xd[r], box[r], HG_S, HG_E)
        assert np.ptp(ref) > 50.0
        assert np.allclose(rows[r], beta * (ref + econst), rtol=rtol), np.abs(rows[r] / (beta * (ref + econst)) - 1).max()
        assert np.isclose(U[r], ref[k], rtol=rtol)
        f_ref = total_energy_forces(desc, xd[r], box[r], HG_S[k], HG_E[k])[1]
        assert np.abs(f[r] - f_ref).max() < ftol * np.abs(f_ref).max(), np.abs(f[r] - f_ref).max() / np.abs(f_ref).max()
    return eng


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(), dict(alchemical_pme_treatment='direct-space')])
def test_the_references_two_region_host_guest_case_on_the_device(hip_engine_factory, kw):
    eng = _check_host_guest(hip_engine_factory(), kw, 1e-5, 2e-4, 'auto')
    assert not np.any(eng.propagate(0))
    assert np.all(np.isfinite(eng.compute_energies()))


def test_the_references_two_region_host_guest_case_on_the_cpu_port():
    if not os.path.exists(CPU_LIB):
        oracle.build()
    _check_host_guest(HipEngine(lib_path=CPU_LIB), dict(), 1e-9, 1e-8, 'reference').close()


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(), dict(alchemical_pme_treatment='direct-space')])
def test_general_regions_under_the_monte_carlo_barostat(hip_engine_factory, kw):
    """NPT states on a System with two regions: the barostat's Metropolis test sees the custom forces (and, under the exact treatment, the
    volume-dependent background term of the regions' scaled charges); after volume moves the u_kl rows -- custom forces at every state's
    lambdas, long-range constants scaled by V_ref / V, + beta p V -- still match the oracle on the device's positions and boxes."""
    al, system, regions = _alanine_two_regions(kw, frozenset())
    nb = [f for f in system.getForces() if isinstance(f, NonbondedForce)][0]
    box0 = np.diag(system.getDefaultPeriodicBoxVectors())
    V0 = float(np.prod(box0))
    econst = alchemy.alchemical_long_range_constants(system, nb, LADDER_S, V0)
    eng = hip_engine_factory()
    desc = system_to_desc(system, ewald_split='auto', min_edge=float(box0.min()) / 1.1)
    eng.set_system(desc)
    K = len(LADDER_S)
    beta = 1.0 / (KB * 300.0)
    p = 1.0 * unit.bar
    eng.set_states(np.full(K, beta), None, None, econst)
    eng.set_region_lambdas(LADDER_S, LADDER_E)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 50, True, 1e-8)
    eng.set_barostat(np.full(K, p), 25)
    eng.set_energy_const_volume(V0)
    eng.seed(5)
    labels = np.array([0, 4, 5])             # (states whose sterics are softened only where the charges are off: the others are for energies, not dynamics)
    x = np.stack([al.positions] * 3)
    eng.set_replicas(3, 0, x, None, np.tile(box0, (3, 1)), labels)
    assert not eng.propagate(0).any()
    boxes = eng.get_boxes()
    assert np.all(np.prod(boxes, axis=1) != V0)
    rows = eng.compute_energies()
    xd = eng.get_replicas()[0]
    for r in range(3):
        V = float(np.prod(boxes[r]))
        ref = total_state_energies(desc, xd[r], boxes[r], LADDER_S, LADDER_E) + econst * V0 / V + p * V
        assert np.allclose(rows[r], beta * ref, rtol=1e-5), np.abs(rows[r] / (beta * ref) - 1).max()


# ---- the factory's default reaction-field treatment: the whole system on an unshifted, switched reaction field -----------------------------
def _check_switched_reaction_field(eng, rtol, ftol):
    """alchemical_rf_treatment='switched' (the default) on a charged fluid: the alchemical atoms' soft-core reaction field with c_rf = 0 and a
    switch (alchemy.py:1473-1508, 1818-1824) AND the environment's pair term re-written the same way (alchemy.py:744-749 ->
    forcefactories.py:76-84 -> forces.UnshiftedReactionFieldForce, forces.py:1110-1150: remd_set_reaction_field)"""
    lj = _charged_lj_fluid()
    plain = ts.LennardJonesFluid(nparticles=216, reduced_density=0.4)
    system = alchemy.AbsoluteAlchemicalFactory(switch_width=0.15).create_alchemical_system(
        lj, alchemy.AlchemicalRegion(alchemical_atoms=range(6), name='lig', softcore_beta=0.3))
    desc = system_to_desc(system)
    assert desc['rf_unshifted_switch_width'] == 0.15 and desc['alch_regions']['elec_crf'] == 0.0 and np.isclose(desc['alch_regions']['elec_switch_distance'], desc['cutoff'] - 0.15)
    nb = [f for f in system.getForces() if isinstance(f, NonbondedForce)][0]
    box0 = np.diag(system.getDefaultPeriodicBoxVectors())
    LS = np.array([[1.0], [0.7], [0.2], [0.0]]); LE = np.array([[1.0], [0.4], [0.0], [0.0]])
    econst = alchemy.alchemical_long_range_constants(system, nb, LS, float(np.prod(box0)))
    eng.set_system(desc)
    beta = 1.0 / (KB * 120.0)
    eng.set_states(np.full(4, beta), None, None, econst)
    eng.set_region_lambdas(LS, LE)
    eng.set_integrator('V R O R V', 0.001, 1.0, 5, True, 1e-8)
    eng.seed(3)
    labels = np.array([0, 2])
    x = np.stack([plain.positions + 0.002 * (r + 1) * np.random.default_rng(r).normal(size=plain.positions.shape) for r in range(2)])
    box = np.tile(box0, (2, 1))
    eng.set_replicas(2, 0, x, None, box, labels)
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    shifted = dict(desc); shifted.pop('rf_unshifted_switch_width')
    for r, k in enumerate(labels):
        ref = total_state_energies(desc, xd[r], box[r], LS, LE)
        assert np.allclose(rows[r], beta * (ref + econst), rtol=rtol, atol=rtol * np.abs(beta * ref).max()), np.abs(rows[r] - beta * (ref + econst)).max()
        assert abs(total_state_energies(shifted, xd[r], box[r], LS, LE)[0] - ref[0]) > 1.0          # OpenMM's shifted field is another Hamiltonian
        f_ref = total_energy_forces(desc, xd[r], box[r], LS[k], LE[k])[1]
        assert np.abs(f[r] - f_ref).max() < ftol * np.abs(f_ref).max()
    return eng


def test_switched_reaction_field_of_the_whole_system_on_the_cpu_port():
    if not os.path.exists(CPU_LIB):
        oracle.build()
    _check_switched_reaction_field(HipEngine(lib_path=CPU_LIB), 1e-9, 1e-8).close()


@pytest.mark.gpu
def test_switched_reaction_field_of_the_whole_system_on_the_device(hip_engine_factory):
    eng = _check_switched_reaction_field(hip_engine_factory(), 2e-5, 2e-4)
    assert not np.any(eng.propagate(0))


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(), dict(alchemical_pme_treatment='direct-space')])
def test_velocity_verlet_conserves_energy_on_a_system_with_regions(hip_engine_factory, kw):
    """forces and energies of the custom-forces launch belong together dynamically: 'V R V' (no thermostat) at 1 fs on solvated alanine
    dipeptide with two regions at intermediate lambdas conserves K + U to a small fraction of kT per degree of freedom over 0.4 ps (the bar of
    test_energy_drift_bounds_the_constraint_solver on the plain system)"""
    al, system, regions = _alanine_two_regions(kw, frozenset(), softcore_beta=0.0 if not kw else 0.25)
    nb = [f for f in system.getForces() if isinstance(f, NonbondedForce)][0]
    box0 = np.diag(system.getDefaultPeriodicBoxVectors())
    eng = hip_engine_factory()
    eng.set_system(system_to_desc(system, ewald_split='auto'))
    LS1, LE1 = np.array([[0.8, 0.7]]), np.array([[0.0, 0.0]]) if kw else np.array([[0.4, 0.0]])
    eng.set_states(np.array([1.0 / (KB * 300.0)]), None, None, alchemy.alchemical_long_range_constants(system, nb, LS1, float(np.prod(box0))))
    eng.set_region_lambdas(LS1, LE1)
    eng.seed(13)
    eng.set_integrator('V R V', 0.001, 0.0, 100, False, 1e-8)
    eng.set_replicas(1, 0, al.positions[None], None, box0[None], np.zeros(1, dtype=int))
    eng.minimize(tolerance=50.0, max_iterations=200)
    eng.set_integrator('V R O R V', 0.001, 5.0, 200, True, 1e-8)
    assert not eng.propagate(0).any()
    eng.set_integrator('V R V', 0.001, 0.0, 100, False, 1e-8)
    energies = []
    for it in range(5):
        kinetic = eng.get_replicas(positions=False, velocities=False, kinetic=True)[3]
        energies.append(float(kinetic[0] + eng.compute_energies(want_potential=True)[1][0]))
        if it < 4:
            assert not eng.propagate(it + 1).any()
    ndof = 3 * 2269 - 2259 - 3
    drift = (np.array(energies) - energies[0]) / (ndof * KB * 300.0)
    assert np.abs(drift).max() < 2e-3, drift
