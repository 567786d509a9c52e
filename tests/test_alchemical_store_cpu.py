"""Compound (alchemical) thermodynamic states in the reference's netCDF4 layout.

The reference stores the System of a CompoundThermodynamicState as the XML of what its AbsoluteAlchemicalFactory built
(multistatereporter.py:612-668, states.py:1257-1280, 2956-2971; alchemy/alchemy.py:1539-2038).  This package marks a region on
a plain System instead, so the store writer has to produce the factory's force set (openmmtools_amd/_alchemical_xml.py).  Checked
here, all on the CPU:

  * the force list, its order, force groups, global parameters, interaction groups and offsets are the factory's;
  * the DOCUMENT means the Hamiltonian the engine evaluates: a small interpreter of the written XML (OpenMM's expression
    syntax, interaction groups, exclusions, cutoff and switching function, parameter offsets) gives the f64 oracle's energy of
    the marked System at several (lambda_sterics, lambda_electrostatics);
  * write -> read returns the description the engine was given; a sampler on alchemical states writes '.nc' files with the
    state dictionaries of CompoundThermodynamicState.__getstate__ / GlobalParameterState.__getstate__ and resumes from them.
"""
import os
import sys
import xml.etree.ElementTree as ET
import zlib

import numpy as np
import pytest
from scipy.special import erfc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from openmmtools_amd import testsystems, alchemy, system_xml, states, mcmc, unit              # noqa: E402
from openmmtools_amd.system import system_to_desc                                             # noqa: E402
from openmmtools_amd.multistate import ReplicaExchangeSampler, MultiStateReporter, _hdf5      # noqa: E402
from oracle.forcefield import ForceFieldOracle                                                # noqa: E402
from oracle_engine import OracleEngine                                                        # noqa: E402

needs_hdf5 = pytest.mark.skipif(not _hdf5.available(), reason='libhdf5 not loadable')


def _alchemical(testsystem, atoms, **region):
    return alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(
        testsystem.system, alchemy.AlchemicalRegion(alchemical_atoms=atoms, **region))


def _forces(xml):
    return list(ET.fromstring(xml).find('Forces'))


def _globals(e):
    return {g.get('name'): float(g.get('default')) for g in e.find('GlobalParameters')}


# ---- structure -----------------------------------------------------------------------------------------------------
def test_pme_system_is_written_as_the_factorys_force_set():
    hg = testsystems.HostGuestExplicit()
    guest = list(range(126, 156))
    xml = system_xml.to_xml(_alchemical(hg, guest))
    f = _forces(xml)
    # alchemy.py:1052-1083: untouched forces, the re-added bonded ones, then 'lambda_electrostatics' (the NonbondedForce with
    # offsets: exact PME treatment :1675-1680) in the lowest free group and the four 'lambda_sterics' forces in the next
    assert [(e.get('type'), e.get('forceGroup')) for e in f] == [
        ('CMMotionRemover', '0'), ('HarmonicBondForce', '0'), ('HarmonicAngleForce', '0'), ('PeriodicTorsionForce', '0'),
        ('NonbondedForce', '1'), ('CustomNonbondedForce', '2'), ('CustomNonbondedForce', '2'), ('CustomBondForce', '2'), ('CustomBondForce', '2')]
    nb, na, aa, na_b, aa_b = f[4:]
    nbf = [x for x in hg.system.getForces() if hasattr(x, 'exceptions')][0]
    assert _globals(nb) == {'lambda_electrostatics': 1.0}
    offs = nb.find('ParticleOffsets').findall('Offset')
    assert [int(o.get('particle')) for o in offs] == guest and all(o.get('parameter') == 'lambda_electrostatics' for o in offs)
    assert np.allclose([float(o.get('q')) for o in offs], [nbf.particles[k][0] for k in guest], rtol=0, atol=0)
    assert all(float(o.get('sig')) == 0.0 and float(o.get('eps')) == 0.0 for o in offs)
    parts = nb.find('Particles').findall('Particle')
    assert all(float(parts[k].get('q')) == 0.0 and float(parts[k].get('eps')) == 0.0 for k in guest)               # :1903-1911
    assert float(parts[0].get('q')) == nbf.particles[0][0] and float(parts[0].get('eps')) == nbf.particles[0][2]
    exc = nb.find('Exceptions').findall('Exception')
    assert len(exc) == len(nbf.exceptions)
    touched = [n for n, (i, j, *_r) in enumerate(nbf.exceptions) if i in guest or j in guest]
    assert all(float(exc[n].get('q')) == 0.0 and float(exc[n].get('eps')) == 0.0 for n in touched)                  # :2000-2006
    eoffs = nb.find('ExceptionOffsets').findall('Offset')
    assert [int(o.get('exception')) for o in eoffs] == [n for n in touched if nbf.exceptions[n][2] != 0.0]           # :1978-1982
    # sterics: lambda-controlled N x A, lambda fixed to 1 for A x A (annihilate_sterics=False) :1767-1779, 1913-1919
    soft = dict(softcore_alpha=0.5, softcore_beta=0.0, softcore_a=1.0, softcore_b=1.0, softcore_c=6.0, softcore_d=1.0, softcore_e=1.0, softcore_f=2.0)
    assert _globals(na) == dict(soft, lambda_sterics=1.0) and _globals(aa) == soft
    assert _globals(na_b) == dict(soft, lambda_sterics=1.0) and _globals(aa_b) == soft
    assert aa.get('energy') == na.get('energy') + 'lambda_sterics=1.0;' and aa_b.get('energy') == na_b.get('energy') + 'lambda_sterics=1.0;'
    assert na.get('energy').startswith('U_sterics;U_sterics = ((lambda_sterics)^softcore_a)*4*epsilon*x*(x-1.0);')
    assert na.get('energy').endswith('epsilon = sqrt(epsilon1*epsilon2);sigma = 0.5*(sigma1 + sigma2);') and 'sigma1' not in na_b.get('energy')
    for e in (na, aa):
        assert [p.get('name') for p in e.find('PerParticleParameters')] == ['sigma', 'epsilon']
        assert len(e.find('Particles')) == 4491 and len(e.find('Exclusions')) == len(nbf.exceptions)
        assert (e.get('method'), float(e.get('cutoff')), e.get('useSwitchingFunction'), e.get('useLongRangeCorrection')) == \
            ('2', nbf.getCutoffDistance(), '1', '1') and float(e.get('switchingDistance')) == nbf.getSwitchingDistance()
    sets = lambda e: [[int(p.get('index')) for p in e.find('InteractionGroups')[0].find(t)] for t in ('Set1', 'Set2')]
    assert sets(na) == [[k for k in range(4491) if k not in guest], guest] and sets(aa) == [guest, guest]
    # the guest's 1-4 Lennard-Jones exceptions are alchemical/alchemical bonds; none straddles the region
    assert len(na_b.find('Bonds')) == 0
    assert len(aa_b.find('Bonds')) == sum(1 for n in touched if nbf.exceptions[n][4] != 0.0) > 0
    assert [p.get('name') for p in aa_b.find('PerBondParameters')] == ['sigma', 'epsilon']


def test_reaction_field_system_gets_the_electrostatics_forces_and_the_unshifted_field():
    lj = testsystems.LennardJonesFluid(nparticles=64)
    f = _forces(system_xml.to_xml(_alchemical(lj, range(4))))
    # no offsets without Ewald: the NonbondedForce stays in group 0, four electrostatics forces (group 1) come before the four
    # sterics ones (group 2) :1799-1830, and replace_reaction_field appends its force (forcefactories.py:76-84)
    assert [(e.get('type'), e.get('forceGroup')) for e in f] == [('NonbondedForce', '0')] + \
        [('CustomNonbondedForce', '1')] * 2 + [('CustomBondForce', '1')] * 2 + [('CustomNonbondedForce', '2')] * 2 + \
        [('CustomBondForce', '2')] * 2 + [('CustomNonbondedForce', '0')]
    na_e, aa_e = f[1], f[2]
    assert 'lambda_electrostatics' in _globals(na_e) and 'lambda_electrostatics' in _globals(aa_e)       # annihilated :1807-1813
    assert na_e.get('energy').startswith('U_electrostatics;U_electrostatics=((lambda_electrostatics)^softcore_d)*ONE_4PI_EPS0*chargeprod'
                                         '*(reff_electrostatics^(-1) + k_rf*reff_electrostatics^2 - c_rf);k_rf = ')
    assert 'c_rf = 0.0;' in na_e.get('energy') and na_e.get('useSwitchingFunction') == '1' and na_e.get('useLongRangeCorrection') == '0'
    assert [p.get('name') for p in na_e.find('PerParticleParameters')] == ['charge', 'sigma']
    rf = f[-1]
    assert rf.get('energy').startswith('ONE_4PI_EPS0*chargeprod*(r^(-1) + k_rf*r^2);chargeprod = charge1*charge2;k_rf = ')
    assert len(rf.find('GlobalParameters')) == 0 and len(rf.find('InteractionGroups')) == 0 and len(rf.find('Particles')) == 64
    # decoupled electrostatics: lambda fixed in the alchemical/alchemical forces
    f = _forces(system_xml.to_xml(_alchemical(lj, range(4), annihilate_electrostatics=False)))
    assert 'lambda_electrostatics' not in _globals(f[2]) and f[2].get('energy').endswith('lambda_electrostatics=1.0;')


def test_what_the_factory_would_build_differently_is_refused():
    al = testsystems.AlanineDipeptideExplicit(nonbondedMethod='CutoffPeriodic')
    # (round 6: a charged region under a reaction-field method is written -- the general force set, the environment's charges in the unshifted
    # reaction-field force, forcefactories.py:76-84 -- and read back)
    rf = _alchemical(al, range(22))
    back, _ = system_xml.from_xml(system_xml.to_xml(rf))
    assert back.alchemical_regions is not None and back.rf_unshifted_switch_width == rf.rf_unshifted_switch_width == 0.1
    assert back.fingerprint() == rf.fingerprint()
    al = testsystems.AlanineDipeptideExplicit()
    with pytest.raises(ValueError, match='Decoupled electrostatics is not supported with exact treatment'):        # alchemy.py:1617-1623
        system_xml.to_xml(_alchemical(al, range(22), annihilate_electrostatics=False))
    xml = system_xml.to_xml(_alchemical(al, range(22)))
    with pytest.raises(NotImplementedError, match='lambda_electrostatics_ligand of a region without sterics forces'):
        system_xml.from_xml(xml.replace('lambda_electrostatics', 'lambda_electrostatics_ligand'))
    with pytest.raises(NotImplementedError, match='outside the alchemical'):
        system_xml.from_xml(xml.replace('U_sterics', 'U_other'))


# ---- meaning: an interpreter of the written document against the oracle -----------------------------------------------
class _Names(dict):
    def __init__(self, values, definitions):
        super().__init__(values)
        self._defs = definitions

    def __missing__(self, name):
        if name not in self._defs:
            raise KeyError(name)
        self[name] = eval(self._defs[name], {'__builtins__': {}}, self)
        return self[name]


def _evaluate(expression, values):
    """OpenMM's expression syntax: 'value; name = expression; ...', ^ for powers."""
    parts = [p.strip() for p in expression.split(';') if p.strip()]
    defs = {}
    for p in parts[1:]:
        name, rhs = p.split('=', 1)
        defs[name.strip()] = rhs.replace('^', '**')
    names = _Names(dict(values, sqrt=np.sqrt, erfc=erfc), defs)
    return eval(parts[0].replace('^', '**'), {'__builtins__': {}}, names)


def _switch(r, rs, rc):
    t = np.clip((r - rs) / (rc - rs), 0.0, 1.0)
    return 1.0 - 10.0 * t ** 3 + 15.0 * t ** 4 - 6.0 * t ** 5


def _document_energy(xml, x, box, lambdas):
    """Energy of an alchemical System document at the given global parameters: custom forces by direct evaluation, everything
    else (with the NonbondedForce's offsets folded into its parameters) through the plain reader and the f64 oracle."""
    root = ET.fromstring(xml)
    forces = root.find('Forces')
    total = 0.0
    for e in list(forces):
        kind = e.get('type')
        if kind == 'NonbondedForce':
            scale = {g.get('name'): lambdas.get(g.get('name'), float(g.get('default'))) for g in e.find('GlobalParameters')}
            parts, excs = e.find('Particles').findall('Particle'), e.find('Exceptions').findall('Exception')
            for o in e.find('ParticleOffsets').findall('Offset'):
                p = parts[int(o.get('particle'))]
                for a in ('q', 'sig', 'eps'):
                    p.set(a, repr(float(p.get(a)) + scale[o.get('parameter')] * float(o.get(a))))
            for o in e.find('ExceptionOffsets').findall('Offset'):
                p = excs[int(o.get('exception'))]
                for a in ('q', 'sig', 'eps'):
                    p.set(a, repr(float(p.get(a)) + scale[o.get('parameter')] * float(o.get(a))))
            for block in ('GlobalParameters', 'ParticleOffsets', 'ExceptionOffsets'):
                for child in list(e.find(block)):
                    e.find(block).remove(child)
        elif kind in ('CustomNonbondedForce', 'CustomBondForce'):
            forces.remove(e)
            g = {p.get('name'): lambdas.get(p.get('name'), float(p.get('default'))) for p in e.find('GlobalParameters')}
            if kind == 'CustomBondForce':
                names = [p.get('name') for p in e.find('PerBondParameters')]
                for b in e.find('Bonds'):
                    d = x[int(b.get('p2'))] - x[int(b.get('p1'))]
                    d -= np.round(d / box) * box * 0          # usesPeriodic = 0
                    vals = dict(g, r=np.linalg.norm(d), **{n: float(b.get('param%d' % (k + 1))) for k, n in enumerate(names)})
                    total += float(_evaluate(e.get('energy'), vals))
                continue
            names = [p.get('name') for p in e.find('PerParticleParameters')]
            par = np.array([[float(p.get('param%d' % (k + 1))) for k in range(len(names))] for p in e.find('Particles')])
            excl = {frozenset((int(p.get('p1')), int(p.get('p2')))) for p in e.find('Exclusions')}
            groups = [[np.array([int(p.get('index')) for p in grp.find(t)]) for t in ('Set1', 'Set2')] for grp in e.find('InteractionGroups')]
            if not groups:
                every = np.arange(len(par))
                groups = [[every, every]]
            rc, rs, sw = float(e.get('cutoff')), float(e.get('switchingDistance')), e.get('useSwitchingFunction') == '1'
            assert e.get('method') == '2'
            for s1, s2 in groups:
                i, j = np.meshgrid(s1, s2, indexing='ij')
                i, j = i.ravel(), j.ravel()
                keep = i != j
                if np.array_equal(s1, s2):
                    keep &= i < j
                i, j = i[keep], j[keep]
                keep = np.array([frozenset((a, b)) not in excl for a, b in zip(i, j)], dtype=bool) if excl else np.ones(len(i), bool)
                i, j = i[keep], j[keep]
                d = x[j] - x[i]
                d -= np.round(d / box) * box
                r = np.linalg.norm(d, axis=1)
                inside = r < rc
                i, j, r = i[inside], j[inside], r[inside]
                vals = dict(g, r=r)
                for k, n in enumerate(names):
                    vals[n + '1'], vals[n + '2'] = par[i, k], par[j, k]
                u = _evaluate(e.get('energy'), vals) * np.ones_like(r)
                total += float((u * (_switch(r, rs, rc) if sw else 1.0)).sum())
    plain, _ = system_xml.from_xml(ET.tostring(root, encoding='unicode'))
    return total + ForceFieldOracle(system_to_desc(plain)).energy_forces(x, box, forces=False)[0]


@pytest.mark.parametrize('name', ['lj-reaction-field', 'alanine-pme', 'alanine-cut-pme', 'host-guest-pme', 'host-guest-annihilate'])
def test_the_document_means_the_hamiltonian_the_engine_evaluates(name):
    if name == 'lj-reaction-field':
        t = testsystems.LennardJonesFluid(nparticles=216, reduced_density=0.6, dispersion_correction=False)
        atoms = range(6)
    elif name == 'alanine-pme':
        t = testsystems.AlanineDipeptideExplicit(use_dispersion_correction=False)
        atoms = range(22)
    elif name == 'alanine-cut-pme':          # the region cuts the solute: 16 soft-core exception bonds, 12 alchemical/alchemical ones
        t = testsystems.AlanineDipeptideExplicit(use_dispersion_correction=False)
        atoms = range(10)
    else:
        t = testsystems.HostGuestExplicit(use_dispersion_correction=False)
        atoms = range(126, 156)
    # annihilate_sterics: the guest's own Lennard-Jones pairs and 1-4 exceptions are lambda-controlled too (alchemy.py:1767-1779, 1841-1846)
    marked = _alchemical(t, atoms, annihilate_sterics=(name == 'host-guest-annihilate'))
    xml = system_xml.to_xml(marked)
    if name == 'host-guest-annihilate':
        aa = [e for e in _forces(xml) if e.get('type') == 'CustomNonbondedForce'][1]
        assert 'lambda_sterics' in _globals(aa) and not aa.get('energy').endswith('lambda_sterics=1.0;')
    box = np.diag(t.system.getDefaultPeriodicBoxVectors())
    x = t.positions + 0.002 * np.random.default_rng(1).normal(size=t.positions.shape)
    oracle = ForceFieldOracle(system_to_desc(marked))
    if name == 'host-guest-annihilate':
        decoupled = ForceFieldOracle(system_to_desc(_alchemical(t, atoms)))
        assert abs(oracle.energy_forces(x, box, lambda_sterics=0.0, forces=False)[0]
                   - decoupled.energy_forces(x, box, lambda_sterics=0.0, forces=False)[0]) > 10.0          # kJ/mol: the guest's own LJ
    for lam_s, lam_e in ((1.0, 1.0), (1.0, 0.35), (0.6, 0.0), (0.0, 0.0)):
        ref = oracle.energy_forces(x, box, lambda_sterics=lam_s, lambda_electrostatics=lam_e, forces=False)[0]
        got = _document_energy(xml, x, box, dict(lambda_sterics=lam_s, lambda_electrostatics=lam_e))
        assert np.isclose(got, ref, rtol=1e-9, atol=1e-6), (name, lam_s, lam_e, got, ref)


# ---- round trip and the store -------------------------------------------------------------------------------------------
@pytest.mark.parametrize('case', ['lj', 'alanine', 'alanine-cut', 'host-guest', 'host-guest-annihilate', 'softcore'])
def test_write_then_read_returns_the_marked_system(case):
    if case == 'lj':
        marked = _alchemical(testsystems.LennardJonesFluid(nparticles=64), range(4))
    elif case == 'alanine':
        marked = _alchemical(testsystems.AlanineDipeptideExplicit(), range(22))
    elif case == 'alanine-cut':
        marked = _alchemical(testsystems.AlanineDipeptideExplicit(), range(10))
    elif case == 'host-guest':
        marked = _alchemical(testsystems.HostGuestExplicit(), range(126, 156))
    elif case == 'host-guest-annihilate':
        marked = _alchemical(testsystems.HostGuestExplicit(), range(126, 156), annihilate_sterics=True)
    else:
        marked = _alchemical(testsystems.LennardJonesFluid(nparticles=64), [3, 9, 11], softcore_alpha=0.3, softcore_a=2, softcore_b=2, softcore_c=6)
        marked.alchemical_lrc = False
    back, barostat = system_xml.from_xml(system_xml.to_xml(marked))
    assert barostat is None and back.alchemical_region.alchemical_atoms == marked.alchemical_region.alchemical_atoms
    r0, r1 = marked.alchemical_region, back.alchemical_region
    assert (r0.softcore_alpha, r0.softcore_a, r0.softcore_b, r0.softcore_c) == (r1.softcore_alpha, r1.softcore_a, r1.softcore_b, r1.softcore_c)
    assert back.alchemical_lrc == marked.alchemical_lrc and r1.annihilate_sterics == r0.annihilate_sterics
    d0, d1 = system_to_desc(marked), system_to_desc(back)
    assert sorted(d0) == sorted(d1)
    for k in d0:
        assert np.array_equal(np.asarray(d0[k]), np.asarray(d1[k])), k
    assert back.fingerprint() == marked.fingerprint()


def _alchemical_sampler(tmp_path, n_iterations, name='alch.nc'):
    lj = testsystems.LennardJonesFluid(nparticles=64)
    asys = _alchemical(lj, range(4))
    ths = [states.CompoundThermodynamicState(states.ThermodynamicState(asys, 120.0 * unit.kelvin),
                                             [states.AlchemicalState(lambda_sterics=l, lambda_electrostatics=1.0)]) for l in (1.0, 0.5, 0.0)]
    ss = states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, n_steps=2, reassign_velocities=True, splitting='V R O R V')
    s = ReplicaExchangeSampler(mcmc_moves=move, number_of_iterations=n_iterations, engine=OracleEngine(ForceFieldOracle), seed=1)
    rep = MultiStateReporter(str(tmp_path / name), checkpoint_interval=1)
    s.create(ths, [ss], storage=rep)
    return s, rep, asys


@needs_hdf5
def test_alchemical_states_go_into_the_netcdf4_layout_and_resume(tmp_path, caplog):
    import logging
    with caplog.at_level(logging.WARNING):
        s, rep, asys = _alchemical_sampler(tmp_path, 3)
    assert not any('record-file container' in r.getMessage() for r in caplog.records)
    s.run()
    rep.close()
    assert os.path.isfile(tmp_path / 'alch.nc') and os.path.isfile(tmp_path / 'alch_checkpoint.nc')
    r = MultiStateReporter(str(tmp_path / 'alch.nc'), open_mode='r')
    # the dictionaries of CompoundThermodynamicState.__getstate__ (states.py:2956-2971) around ThermodynamicState's (:1257-1280)
    # and GlobalParameterState's (:3879-3898); the System only in the first state of the compatible group (:640-662)
    d0, d2 = r.read_dict('thermodynamic_states/state0'), r.read_dict('thermodynamic_states/state2')
    for d in (d0, d2):
        assert (d['_serialized__class_name'], d['_serialized__module_name']) == ('CompoundThermodynamicState', 'openmmtools.states')
        assert sorted(d) == ['_serialized__class_name', '_serialized__module_name', 'composable_states', 'thermodynamic_state']
        (c,) = d['composable_states']
        assert (c['_serialized__class_name'], c['_serialized__module_name']) == ('AlchemicalState', 'openmmtools.alchemy.alchemy')
        assert c['function_variables'] == {} and c['parameters_name_suffix'] is None
        assert sorted(c['parameters']) == ['lambda_angles', 'lambda_bonds', 'lambda_electrostatics', 'lambda_sterics', 'lambda_torsions']
        inner = d['thermodynamic_state']
        assert (inner['_serialized__class_name'], inner['_serialized__module_name']) == ('ThermodynamicState', 'openmmtools.states')
        assert inner['temperature'] == 120.0 and inner['pressure'] is None and inner['surface_tension'] is None
    assert d0['composable_states'][0]['parameters'] == dict(lambda_sterics=1.0, lambda_electrostatics=1.0, lambda_bonds=None,
                                                            lambda_angles=None, lambda_torsions=None)
    assert d2['composable_states'][0]['parameters']['lambda_sterics'] == 0.0
    assert 'standard_system' in d0['thermodynamic_state'] and '_Reporter__compatible_state' not in d0['thermodynamic_state']
    assert d2['thermodynamic_state']['_Reporter__compatible_state'] == 'thermodynamic_states/0' and 'standard_system' not in d2['thermodynamic_state']
    xml = zlib.decompress(d0['thermodynamic_state']['standard_system']).decode()
    assert xml == system_xml.to_xml(asys)
    th, un = r.read_thermodynamic_states()
    assert [type(t).__name__ for t in th] == ['CompoundThermodynamicState'] * 3 and un == []
    assert [t.lambda_sterics for t in th] == [1.0, 0.5, 0.0] and all(t.lambda_electrostatics == 1.0 for t in th)
    assert th[1].system is th[0].system and th[0].system.fingerprint() == asys.fingerprint()
    e_first = r.read_energies()[0]
    r.close()
    # the HDF5 objects are those of a store the reference wrote (the shipped legacy store; h5dump is optional)
    import shutil
    if shutil.which('h5dump'):
        from test_reference_store import _h5_structure, STORE
        ours, theirs = _h5_structure(str(tmp_path / 'alch.nc')), _h5_structure(STORE)
        for var in ('/thermodynamic_states/state0', '/thermodynamic_states/state1'):
            assert ours[var]['type'] == theirs[var]['type']
            assert [d.rstrip('0123456789') for d in ours[var]['dims']] == [d.rstrip('0123456789') for d in theirs[var]['dims']] == ['/fixedL']
    # resume in place, against an uninterrupted run
    full, rep_full, _ = _alchemical_sampler(tmp_path, 5, name='full.nc')
    full.run()
    rep_full.close()
    res = ReplicaExchangeSampler.from_storage(str(tmp_path / 'alch.nc'), engine=OracleEngine(ForceFieldOracle))
    assert res.iteration == 3
    res.extend(2)
    res._reporter.close()
    ea = MultiStateReporter(str(tmp_path / 'alch.nc'), open_mode='r').read_energies()[0]
    eb = MultiStateReporter(str(tmp_path / 'full.nc'), open_mode='r').read_energies()[0]
    assert ea.shape == eb.shape == (6, 3, 3) and np.array_equal(ea[:4], e_first[:4])
    assert np.array_equal(ea[:4], eb[:4]) and np.allclose(ea[4:], eb[4:], rtol=2e-5, atol=1e-6)        # restart from f4 checkpoints


def test_a_document_in_the_dress_of_an_older_openmm_is_read():
    """The reader is tolerant of what differs between OpenMM releases (no `name` attribute and version 1 / 2 forces in 7.x, no
    ComputedValues / EnergyParameterDerivatives blocks, `true` / `false` is not used by OpenMM but harmless, attribute order free):
    a hand-written alchemical System of two alchemical and two environment atoms, with one soft-core exception, parses into the
    marked System it describes."""
    from openmmtools_amd import _alchemical_xml as ax
    pair = ax.sterics_exception_expression() + 'epsilon = sqrt(epsilon1*epsilon2);sigma = 0.5*(sigma1 + sigma2);'
    soft = ''.join('<Parameter default="%s" name="%s"/>' % (v, n) for n, v in
                   (('softcore_alpha', 0.5), ('softcore_beta', 0.0), ('softcore_a', 1.0), ('softcore_b', 1.0), ('softcore_c', 6.0),
                    ('softcore_d', 1.0), ('softcore_e', 1.0), ('softcore_f', 2.0)))
    particles = ''.join('<Particle param1="%s" param2="%s"/>' % p for p in ((0.3, 0.5), (0.25, 0.0), (0.32, 0.6), (0.31, 0.4)))
    cnb = lambda lam, s1, s2, energy: (
        '<Force cutoff="1" energy="%s" forceGroup="2" method="2" switchingDistance="0.9" type="CustomNonbondedForce" useLongRangeCorrection="1" '
        'useSwitchingFunction="1" version="2"><PerParticleParameters><Parameter name="sigma"/><Parameter name="epsilon"/></PerParticleParameters>'
        '<GlobalParameters>%s%s</GlobalParameters><Particles>%s</Particles><Exclusions><Exclusion p1="1" p2="2"/></Exclusions><Functions/>'
        '<InteractionGroups><InteractionGroup><Set1>%s</Set1><Set2>%s</Set2></InteractionGroup></InteractionGroups></Force>'
        % (energy, lam, soft, particles, ''.join('<Particle index="%d"/>' % i for i in s1), ''.join('<Particle index="%d"/>' % i for i in s2)))
    lam = '<Parameter default="1" name="lambda_sterics"/>'
    bond = lambda lam, energy, bonds: (
        '<Force energy="%s" forceGroup="2" type="CustomBondForce" usesPeriodic="0" version="1"><PerBondParameters><Parameter name="sigma"/>'
        '<Parameter name="epsilon"/></PerBondParameters><GlobalParameters>%s%s</GlobalParameters><Bonds>%s</Bonds></Force>' % (energy, lam, soft, bonds))
    doc = ('<?xml version="1.0" ?><System openmmVersion="7.7" type="System" version="1">'
           '<PeriodicBoxVectors><A x="3" y="0" z="0"/><B x="0" y="3" z="0"/><C x="0" y="0" z="3"/></PeriodicBoxVectors>'
           '<Particles><Particle mass="12"/><Particle mass="1"/><Particle mass="16"/><Particle mass="14"/></Particles><Constraints/><Forces>'
           '<Force alpha="0" cutoff="1" dispersionCorrection="1" ewaldTolerance=".0005" forceGroup="1" method="4" nx="0" ny="0" nz="0" '
           'recipForceGroup="-1" rfDielectric="78.3" switchingDistance="0.9" type="NonbondedForce" useSwitchingFunction="1" version="3">'
           '<GlobalParameters><Parameter default="1" name="lambda_electrostatics"/></GlobalParameters>'
           '<ParticleOffsets><Offset eps="0" parameter="lambda_electrostatics" particle="0" q="-0.2" sig="0"/>'
           '<Offset eps="0" parameter="lambda_electrostatics" particle="1" q="0.2" sig="0"/></ParticleOffsets>'
           '<ExceptionOffsets><Offset eps="0" exception="0" parameter="lambda_electrostatics" q="-0.01" sig="0"/></ExceptionOffsets>'
           '<Particles><Particle eps="0" q="0" sig=".3"/><Particle eps="0" q="0" sig=".25"/><Particle eps=".6" q="-.4" sig=".32"/>'
           '<Particle eps=".4" q=".4" sig=".31"/></Particles><Exceptions><Exception eps="0" p1="1" p2="2" q="0" sig=".285"/></Exceptions></Force>'
           + cnb(lam, (2, 3), (0, 1), pair) + cnb('', (0, 1), (0, 1), pair + 'lambda_sterics=1.0;')
           + bond(lam, ax.sterics_exception_expression(), '<Bond p1="1" p2="2" param1=".285" param2=".05"/>')
           + bond('', ax.sterics_exception_expression() + 'lambda_sterics=1.0;', '')
           + '</Forces></System>')
    system, barostat = system_xml.from_xml(doc)
    assert barostat is None and system.alchemical_region.alchemical_atoms == [0, 1] and not system.alchemical_region.annihilate_sterics
    assert system.alchemical_lrc is True
    nb = [f for f in system.getForces() if hasattr(f, 'exceptions')][0]
    assert nb.particles == [(-0.2, 0.3, 0.5), (0.2, 0.25, 0.0), (-0.4, 0.32, 0.6), (0.4, 0.31, 0.4)]
    assert nb.exceptions == [(1, 2, -0.01, 0.285, 0.05)] and nb.getForceGroup() == 0
    d = system_to_desc(system)
    assert list(d['alch_atoms']) == [0, 1] and d['annihilate_sterics'] is False and d['nb_method'] == 2
    # and what is written from it reads back the same
    again, _ = system_xml.from_xml(system_xml.to_xml(system))
    assert again.fingerprint() == system.fingerprint()
