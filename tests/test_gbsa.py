"""GBSA implicit solvent (OBC2 + ACE) of NoCutoff systems: openmm.GBSAOBCForce in the form the reference's alchemical factory spells out
(/root/reference/openmmtools/alchemy/alchemy.py:2144-2225, _alchemically_modify_GBSAOBCForce), alchemical and not.

  * oracle/gbsa.py (f64) reproduces the VALUES of the reference's own CustomGBForce expression strings -- computed values I, B and the energy --
    on random small systems at three lambda (tests/golden/reference_gbsa.json, made by tests/golden/make_golden_gbsa.py from the reference's
    syntax tree + an interpreter of the CustomGBForce semantics);
  * the C++ build of the ABI (analytic forces through the Born radii) and -- under -m gpu -- the HIP kernels (csrc/gbsa.hip) against that
    oracle (autograd forces) on the implicit-solvent dipeptide: potential, u_kl over a lambda ladder, forces, a short propagation.
"""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from oracle.forcefield import ForceFieldOracle
from oracle.gbsa import gbsa_energy_torch, gbsa_energy_forces
from oracle.alchemical_regions import total_state_energies, total_energy_forces
from openmmtools_amd import alchemy, states, mcmc, unit, system_xml, testsystems as ts
from openmmtools_amd.system import system_to_desc, GBSAOBCForce
from openmmtools_amd._engine import HipEngine

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, 'golden', 'reference_gbsa.json')))
KB = 0.008314462618153242
CPU_LIB = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')


def test_oracle_reproduces_the_references_custom_gb_expressions():
    assert len(G['cases']) == 6 and [c[0] for c in G['computed_values']] == ['I', 'B'] and len(G['energy_terms']) == 3
    for c in G['cases']:
        e, I, B = gbsa_energy_torch(torch.tensor(c['x'], dtype=torch.float64), c['charge'], c['radius'], c['scale'], c['alchemical'],
                                    c['lambda_electrostatics'], c['soluteDielectric'], c['solventDielectric'], return_parts=True)
        assert np.isclose(float(e), c['energy'], rtol=1e-13) and np.allclose(I.numpy(), c['I'], rtol=1e-13) and np.allclose(B.numpy(), c['B'], rtol=1e-13)
    assert G['globals_in_the_function'] == {'lambda_electrostatics': 1.0, 'offset': 0.009}


def _gb_total(desc, x, le, forces=True):
    d0 = dict(desc)
    gb = d0.pop('gbsa')
    e1, f1 = gbsa_energy_forces(x, gb['charge'], gb['radius'], gb['scale'], gb['alchemical'], le, gb['solute_dielectric'], gb['solvent_dielectric'],
                                sasa=bool(gb['surface_area']), forces=forces)
    return d0, e1, f1


def _check_plain(eng, rtol, ftol):
    al = ts.AlanineDipeptideImplicit()
    assert sum(isinstance(f, GBSAOBCForce) for f in al.system.getForces()) == 1
    desc = system_to_desc(al.system)
    assert desc['nb_method'] == 3 and desc['gbsa']['surface_area'] == 1 and np.all(desc['gbsa']['alchemical'] == 0)
    eng.set_system(desc)
    T = np.array([300.0, 350.0])
    eng.set_states(1.0 / (KB * T))
    eng.set_integrator('V R O R V', 0.002, 1.0, 25, True, 1e-8)
    eng.seed(6)
    x = np.stack([al.positions + 0.003 * (r + 1) * np.random.default_rng(r).normal(size=al.positions.shape) for r in range(2)])
    eng.set_replicas(2, 0, x, None, np.zeros((2, 3)), np.arange(2))
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    for r in range(2):
        d0, e1, f1 = _gb_total(desc, xd[r], 1.0)
        e0, f0 = ForceFieldOracle(d0).energy_forces(xd[r], None)
        assert e1 < -20.0                                            # tens of kJ/mol of solvation
        assert np.isclose(U[r], e0 + e1, rtol=rtol, atol=rtol * 100.0), (U[r], e0 + e1)
        assert np.allclose(rows[r], (e0 + e1) / (KB * T), rtol=rtol, atol=rtol * 100.0)
        assert np.abs(f[r] - (f0 + f1)).max() < ftol * np.abs(f0 + f1).max()
    assert not np.any(eng.propagate(0))
    assert np.all(np.isfinite(eng.compute_energies()))
    return eng


def _droplet_in_implicit_solvent(n=150):
    """a droplet of charged Lennard-Jones particles with GBSA parameters: more than two 64-atom tiles, an alchemical half (the tiled GB
    kernels and the multi-wavefront NoCutoff sum beyond one tile; the dipeptide has 22 atoms)"""
    from openmmtools_amd.system import System, NonbondedForce
    g = np.stack(np.meshgrid(*[np.arange(6)] * 3, indexing='ij'), axis=-1).reshape(-1, 3) * 0.38
    x = g[np.argsort(np.linalg.norm(g - g.mean(0), axis=1), kind='stable')[:n]].astype(np.float64)
    s = System()
    nb = NonbondedForce(); nb.setNonbondedMethod(NonbondedForce.NoCutoff)
    gb = GBSAOBCForce()
    rng = np.random.default_rng(5)
    for i in range(n):
        s.addParticle(39.9)
        q = 0.3 if i % 2 == 0 else -0.3
        nb.addParticle(q, 0.34, 0.5)
        gb.addParticle(q, 0.15 + 0.05 * rng.random(), 0.7 + 0.2 * rng.random())
    nb.addException(0, 1, 0.0, 0.3, 0.0)
    s.addForce(nb); s.addForce(gb)
    return s, x


def _check_droplet(eng, rtol, ftol):
    system, x0 = _droplet_in_implicit_solvent()
    desc = system_to_desc(system)
    eng.set_system(desc)
    T = np.array([300.0, 350.0])
    eng.set_states(1.0 / (KB * T))
    eng.set_integrator('V R O R V', 0.001, 1.0, 10, True, 1e-8)
    eng.seed(6)
    x = np.stack([x0 + 0.004 * (r + 1) * np.random.default_rng(r).normal(size=x0.shape) for r in range(2)])
    eng.set_replicas(2, 0, x, None, np.zeros((2, 3)), np.arange(2))
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    for r in range(2):
        d0, e1, f1 = _gb_total(desc, xd[r], 1.0)
        e0, f0 = ForceFieldOracle(d0).energy_forces(xd[r], None)
        assert np.isclose(U[r], e0 + e1, rtol=rtol, atol=rtol * 100.0), (U[r], e0 + e1)
        assert np.abs(f[r] - (f0 + f1)).max() < ftol * np.abs(f0 + f1).max()
    assert not np.any(eng.propagate(0))


def test_cpu_port_evaluates_a_droplet_of_150_atoms_in_implicit_solvent_like_the_oracle():
    if not os.path.exists(CPU_LIB):
        pytest.skip('oracle/_build/libremd_cpu.so not built (make -C oracle)')
    _check_droplet(HipEngine(lib_path=CPU_LIB), 1e-9, 1e-8)


@pytest.mark.gpu
def test_hip_evaluates_a_droplet_of_150_atoms_in_implicit_solvent_like_the_oracle(hip_engine_factory):
    _check_droplet(hip_engine_factory(), 5e-6, 2e-4)


LS = np.array([[1.0], [1.0], [0.5], [0.0]])
LE = np.array([[1.0], [0.4], [0.0], [0.0]])


def _check_alchemical(eng, rtol, ftol):
    """the factory on an implicit-solvent System: the NonbondedForce's custom forces (general-regions path, NoCutoff) and the GB terms with
    lambda_electrostatics on the alchemical particles (alchemy.py:2195-2210)"""
    al = ts.AlanineDipeptideImplicit()
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(6)))
    desc = system_to_desc(system)
    assert desc['gbsa']['alchemical'].tolist() == [1] * 6 + [0] * 16 and desc['alch_regions']['electrostatics'] == 1
    eng.set_system(desc)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(4, beta))
    eng.set_region_lambdas(LS, LE)
    eng.set_integrator('V R O R V', 0.002, 1.0, 10, True, 1e-8)
    eng.seed(8)
    labels = np.array([1, 2, 3])
    x = np.stack([al.positions + 0.003 * (r + 1) * np.random.default_rng(r).normal(size=al.positions.shape) for r in range(3)])
    eng.set_replicas(3, 0, x, None, np.zeros((3, 3)), labels)
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    for r, k in enumerate(labels):
        d0 = dict(desc); d0.pop('gbsa')
        ref = total_state_energies(d0, xd[r], None, LS, LE) + np.array([_gb_total(desc, xd[r], LE[q, 0], forces=False)[1] for q in range(4)])
        assert np.ptp(ref) > 10.0
        assert np.allclose(rows[r], beta * ref, rtol=rtol, atol=rtol * np.abs(beta * ref).max()), np.abs(rows[r] - beta * ref).max()
        assert np.isclose(U[r], ref[k], rtol=rtol, atol=rtol * np.abs(ref).max())
        f_ref = total_energy_forces(d0, xd[r], None, LS[k], LE[k])[1] + _gb_total(desc, xd[r], LE[k, 0])[2]
        assert np.abs(f[r] - f_ref).max() < ftol * np.abs(f_ref).max()
    assert not np.any(eng.propagate(0))
    return eng


def test_cpu_port_evaluates_the_implicit_solvent_dipeptide_like_the_oracle():
    if not os.path.exists(CPU_LIB):
        oracle.build()
    _check_plain(HipEngine(lib_path=CPU_LIB), 1e-10, 1e-9).close()
    _check_alchemical(HipEngine(lib_path=CPU_LIB), 1e-10, 1e-9).close()


# REMD_GB_SMALL: systems of up to 64 atoms take ONE launch (gb_small_kernel: 16 wavefronts per replica share an atom's partners); '0' keeps
# them on the three launches of the general path -- both against the oracle
@pytest.mark.gpu
@pytest.mark.parametrize('small', ['1', '0'])
def test_hip_evaluates_the_implicit_solvent_dipeptide_like_the_oracle(hip_engine_factory, monkeypatch, small):
    monkeypatch.setenv('REMD_GB_SMALL', small)
    _check_plain(hip_engine_factory(), 5e-6, 2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('small', ['1', '0'])
def test_hip_alchemical_implicit_solvent_dipeptide(hip_engine_factory, monkeypatch, small):
    monkeypatch.setenv('REMD_GB_SMALL', small)
    _check_alchemical(hip_engine_factory(), 1e-5, 2e-4)


def test_gbsa_force_in_a_system_document_and_what_is_refused():
    al = ts.AlanineDipeptideImplicit()
    back, _ = system_xml.from_xml(system_xml.to_xml(al.system))
    assert back.fingerprint() == al.system.fingerprint()
    with pytest.raises(NotImplementedError, match='OBC2'):
        ts.AlanineDipeptideImplicit(implicitSolvent='OBC1')
    two = [alchemy.AlchemicalRegion(alchemical_atoms=range(6), name='a'), alchemy.AlchemicalRegion(alchemical_atoms=range(6, 16), name='b')]
    with pytest.raises(NotImplementedError, match='Multiple regions does not work with GBSAOBCForce'):          # alchemy.py:2168-2169
        alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(al.system, two)
    # the alchemical System: the factory's CustomGBForce with the reference's strings, verbatim, in the lambda_electrostatics force group
    from openmmtools_amd import _alchemical_xml as ax
    assert [ax._GB_I, ax._GB_B] == [v[1] for v in G['computed_values']] and [ax._GB_SELF, ax._GB_SURFACE, ax._GB_PAIR] == [t[0] for t in G['energy_terms']]
    assert [v[2] for v in G['computed_values']] == ['ParticlePairNoExclusions', 'SingleParticle']
    marked = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(6)))
    xml = system_xml.to_xml(marked)
    import xml.etree.ElementTree as ET
    forces = ET.fromstring(xml).find('Forces').findall('Force')
    cgb = [f for f in forces if f.get('type') == 'CustomGBForce']
    assert len(cgb) == 1 and not [f for f in forces if f.get('type') == 'GBSAOBCForce']
    elec_group = [f.get('forceGroup') for f in forces if 'U_electrostatics' in f.get('energy', '')]
    assert cgb[0].get('forceGroup') == elec_group[0] and [p.get('param4') for p in cgb[0].find('Particles')][:7] == ['1.0'] * 6 + ['0.0']
    back, _ = system_xml.from_xml(xml)
    assert back.fingerprint() == marked.fingerprint()


def _alchemical_implicit_sampler(engine, n_iterations):
    from openmmtools_amd.multistate import ReplicaExchangeSampler
    al = ts.AlanineDipeptideImplicit()
    asys = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(6)))
    ths = [states.CompoundThermodynamicState(states.ThermodynamicState(asys, 300.0 * unit.kelvin),
                                             [states.AlchemicalState(lambda_sterics=float(ls), lambda_electrostatics=float(le))]) for ls, le in zip(LS[:, 0], LE[:, 0])]
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, n_steps=10, reassign_velocities=True, splitting='V R O R V')
    s = ReplicaExchangeSampler(mcmc_moves=move, number_of_iterations=n_iterations, engine=engine, seed=12, online_analysis_interval=None)
    s.create(ths, [states.SamplerState(al.positions)], storage=None)
    return s, asys


def _check_sampler(s, asys, rtol):
    s._compute_energies()
    desc = system_to_desc(asys)
    x = s._engine.get_replicas()[0]
    d0 = dict(desc); d0.pop('gbsa')
    for r in range(4):
        ref = total_state_energies(d0, x[r], None, LS, LE) + np.array([_gb_total(desc, x[r], LE[q, 0], forces=False)[1] for q in range(4)])
        ref = ref / (KB * 300.0)
        assert np.allclose(s.energy_thermodynamic_states[r], ref, rtol=rtol, atol=rtol * np.abs(ref).max())
    s.run()
    assert np.all(np.isfinite(s.energy_thermodynamic_states))


def test_replica_exchange_over_an_alchemical_ladder_in_implicit_solvent_on_the_cpu_port():
    """the sampler end to end on an alchemical GBSA System: one unsuffixed AlchemicalState per state reaches the custom forces AND the GB
    terms (region 1's lambda_electrostatics), u_kl of the first energy pass against the oracle, then iterations with swaps"""
    if not os.path.exists(CPU_LIB):
        oracle.build()
    s, asys = _alchemical_implicit_sampler(HipEngine(lib_path=CPU_LIB), 3)
    _check_sampler(s, asys, 1e-9)
    assert s.iteration == 3


@pytest.mark.gpu
def test_replica_exchange_over_an_alchemical_ladder_in_implicit_solvent_on_the_device(hip_engine_factory):
    s, asys = _alchemical_implicit_sampler(hip_engine_factory(), 4)
    _check_sampler(s, asys, 2e-5)
    assert s.iteration == 4


@pytest.mark.gpu
def test_velocity_verlet_conserves_energy_in_implicit_solvent(hip_engine_factory):
    """the three GB launches' forces belong to the GB energy dynamically: 'V R V' at 1 fs on the implicit-solvent dipeptide conserves K + U
    to a small fraction of kT per degree of freedom over 2 ps"""
    al = ts.AlanineDipeptideImplicit()
    eng = hip_engine_factory()
    eng.set_system(system_to_desc(al.system))
    eng.set_states(np.array([1.0 / (KB * 300.0)]))
    eng.seed(3)
    eng.set_integrator('V R O R V', 0.001, 5.0, 500, True, 1e-8)
    eng.set_replicas(1, 0, al.positions[None], None, np.zeros((1, 3)), np.zeros(1, dtype=int))
    assert not eng.propagate(0).any()
    eng.set_integrator('V R V', 0.001, 0.0, 500, False, 1e-8)
    energies = []
    for it in range(5):
        kinetic = eng.get_replicas(positions=False, velocities=False, kinetic=True)[3]
        energies.append(float(kinetic[0] + eng.compute_energies(want_potential=True)[1][0]))
        if it < 4:
            assert not eng.propagate(it + 1).any()
    ndof = 3 * 22 - 12 - 3
    drift = (np.array(energies) - energies[0]) / (ndof * KB * 300.0)
    assert np.abs(drift).max() < 5e-3, drift
