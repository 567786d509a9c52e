"""NonbondedForce with NoCutoff: the reference's vacuum test systems (testsystems.AlanineDipeptideVacuum, testsystems.py:3352-3388;
openmm.NonbondedForce.NoCutoff) -- every pair, plain Lennard-Jones + Coulomb, no box, no switch, no dispersion correction; and their
alchemical versions, which always take the general-regions path (the custom forces copy the NonbondedForce's method, alchemy.py:1793-1796;
soft-core Coulomb l^d qq / r_eff, :1434-1447).

The f64 oracle (oracle/forcefield.py method 3 + oracle/alchemical_regions.py) against the C++ build of the ABI here and the HIP kernels
(csrc/nocutoff.hip, csrc/alch_regions.hip) under -m gpu; the soft-core Coulomb of the NoCutoff method is pinned to the reference's own
expression string (tests/golden/reference_alchemy_expressions.json: electrostatics_nocutoff).
"""
import copy
import json
import os

import numpy as np
import pytest

import oracle
from oracle.forcefield import ForceFieldOracle
from oracle.alchemical_regions import RegionOracle, total_state_energies, total_energy_forces
from openmmtools_amd import alchemy, states, mcmc, unit, testsystems as ts
from openmmtools_amd.system import System, NonbondedForce, system_to_desc
from openmmtools_amd._engine import HipEngine
from openmmtools_amd.multistate import ParallelTemperingSampler

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, 'golden', 'reference_alchemy_expressions.json')))
KB = 0.008314462618153242
CPU_LIB = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')


def _cluster(n=150):
    """a droplet of charged Lennard-Jones particles without a box (more than two 64-atom tiles; some pairs excluded, some excepted)"""
    lj = ts.LennardJonesFluid(nparticles=216, reduced_density=0.6)
    g = np.stack(np.meshgrid(*[np.arange(6)] * 3, indexing='ij'), axis=-1).reshape(-1, 3) * 0.38          # a lattice at the Lennard-Jones minimum
    x = g[np.argsort(np.linalg.norm(g - g.mean(0), axis=1), kind='stable')[:n]].astype(np.float64)
    s = System()
    nb = NonbondedForce()
    nb.setNonbondedMethod(NonbondedForce.NoCutoff)
    nb0 = [f for f in lj.system.getForces() if isinstance(f, NonbondedForce)][0]
    for i in range(n):
        s.addParticle(39.9)
        nb.addParticle(0.2 if i % 2 == 0 else -0.2, nb0.particles[i][1], nb0.particles[i][2])
    nb.addException(0, 1, 0.0, 0.3, 0.0)
    nb.addException(5, min(70, n - 2), -0.01, 0.33, 0.4)
    nb.addException(3, min(140, n - 1), 0.02, 0.3, 0.0)
    s.addForce(nb)
    return s, x


def _check_plain(eng, system, x0, rtol, ftol, R=2):
    desc = system_to_desc(system)
    assert desc['nb_method'] == 3 and desc['use_dispersion_correction'] == 0
    eng.set_system(desc)
    T = np.array([300.0, 400.0])
    eng.set_states(1.0 / (KB * T))
    eng.set_integrator('V R O R V', 0.001, 1.0, 20, True, 1e-8)
    eng.seed(9)
    x = np.stack([x0 + 0.002 * (r + 1) * np.random.default_rng(r).normal(size=x0.shape) for r in range(R)])
    eng.set_replicas(R, 0, x, None, np.zeros((R, 3)), np.arange(R))
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    ff = ForceFieldOracle(desc)
    for r in range(R):
        e_ref, f_ref = ff.energy_forces(xd[r], None)
        assert np.isclose(U[r], e_ref, rtol=rtol, atol=rtol * 10.0), (U[r], e_ref)
        assert np.allclose(rows[r], e_ref / (KB * T), rtol=rtol, atol=rtol * 10.0)
        assert np.abs(f[r] - f_ref).max() < ftol * np.abs(f_ref).max()
    return desc


def _system(which):
    if which == 'cluster':
        return _cluster()
    a = ts.AlanineDipeptideVacuum() if which == 'alanine' else ts.HostGuestVacuum()          # testsystems.py:3352-3388, 3660-3712
    return a.system, a.positions


@pytest.mark.parametrize('which', ['alanine', 'cluster', 'hostguest'])
def test_cpu_port_evaluates_nocutoff_systems_like_the_oracle(which):
    if not os.path.exists(CPU_LIB):
        oracle.build()
    system, x = _system(which)
    eng = HipEngine(lib_path=CPU_LIB)
    _check_plain(eng, system, x, 1e-10, 1e-9)
    assert not np.any(eng.propagate(0))
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('which', ['alanine', 'cluster', 'hostguest'])
def test_hip_evaluates_nocutoff_systems_like_the_oracle(hip_engine_factory, which):
    system, x = _system(which)
    eng = hip_engine_factory()
    _check_plain(eng, system, x, 2e-6, 1e-4)
    assert not np.any(eng.propagate(0))
    assert np.all(np.isfinite(eng.compute_energies()))


def test_vacuum_system_shape_and_soft_core_coulomb_of_the_nocutoff_method():
    al = ts.AlanineDipeptideVacuum()
    assert al.system.getNumParticles() == 22 and not al.system.usesPeriodicBoundaryConditions() and al.system.getNumConstraints() == 12
    # the alchemical version: always the general-regions path; its electrostatics the reference's NoCutoff expression (alchemy.py:1434-1447)
    asys = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(6)))
    assert asys.alchemical_regions is not None and asys.alchemical_region is None
    t = system_to_desc(asys)['alch_regions']
    assert t['electrostatics'] == 1 and t['elec_alpha'] == 0.0 and t['elec_krf'] == 0.0 and t['elec_switch_distance'] == -1.0 and t['exact_pme'] == 0
    n = 0
    for smp in G['samples']['electrostatics_nocutoff']:
        s = System()
        s.addParticle(12.0); s.addParticle(12.0)
        nb = NonbondedForce()
        nb.setNonbondedMethod(NonbondedForce.NoCutoff)
        nb.addParticle(smp['charge1'], smp['sigma1'], 0.0); nb.addParticle(smp['charge2'], smp['sigma2'], 0.0)
        s.addForce(nb)
        a = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(s, alchemy.AlchemicalRegion(alchemical_atoms=[0], softcore_beta=smp['softcore_beta']))
        d = system_to_desc(a)
        reg = RegionOracle(d['alch_regions'], d['cutoff'], None, np.zeros((0, 2), int))
        x = np.array([[0.0, 0.0, 0.0], [0.6 * smp['r'], -0.48 * smp['r'], 0.64 * smp['r']]])
        got = reg.energy_forces(x, None, [1.0], [smp['lambda_electrostatics']], forces=False)[0]
        assert np.isclose(got, smp['value'], rtol=1e-12), (smp, got)
        n += 1
    assert n == 40


LS = np.array([[1.0], [1.0], [0.6], [0.0]])
LE = np.array([[1.0], [0.3], [0.0], [0.0]])


def _check_alchemical(eng, rtol, ftol):
    al = ts.AlanineDipeptideVacuum()
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(
        al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(6), softcore_beta=0.2, alchemical_torsions=True))
    desc = system_to_desc(system)
    eng.set_system(desc)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(4, beta))
    eng.set_region_lambdas(LS, LE)
    BONDED = np.ones((4, 3, 1)); BONDED[:, 2, 0] = [1.0, 0.8, 0.4, 0.0]
    eng.set_region_bonded_lambdas(None, None, BONDED[:, 2])
    eng.set_integrator('V R O R V', 0.001, 1.0, 10, True, 1e-8)
    eng.seed(2)
    labels = np.array([0, 2, 3])
    x = np.stack([al.positions + 0.002 * (r + 1) * np.random.default_rng(r).normal(size=al.positions.shape) for r in range(3)])
    eng.set_replicas(3, 0, x, None, np.zeros((3, 3)), labels)
    rows, U = eng.compute_energies(want_potential=True)
    xd = eng.get_replicas()[0]
    f = eng.get_forces()
    for r, k in enumerate(labels):
        ref = total_state_energies(desc, xd[r], None, LS, LE, BONDED)
        assert np.ptp(ref) > 10.0
        assert np.allclose(rows[r], beta * ref, rtol=rtol, atol=rtol * np.abs(beta * ref).max())
        assert np.isclose(U[r], ref[k], rtol=rtol, atol=rtol * np.abs(ref).max())
        f_ref = total_energy_forces(desc, xd[r], None, LS[k], LE[k], tuple(BONDED[k]))[1]
        assert np.abs(f[r] - f_ref).max() < ftol * np.abs(f_ref).max()
    assert not np.any(eng.propagate(0))
    return eng


def test_alchemical_vacuum_system_on_the_cpu_port():
    if not os.path.exists(CPU_LIB):
        oracle.build()
    _check_alchemical(HipEngine(lib_path=CPU_LIB), 1e-10, 1e-9).close()


@pytest.mark.gpu
def test_alchemical_vacuum_system_on_the_device(hip_engine_factory):
    _check_alchemical(hip_engine_factory(), 5e-6, 2e-4)


@pytest.mark.gpu
def test_parallel_tempering_of_the_vacuum_dipeptide_on_the_device(hip_engine_factory):
    """the sampler on a NoCutoff system: four temperatures, swaps, the potential of every replica finite and bounded (a 22-atom molecule
    does not explode), temperature scaling of the rows exact (paralleltempering.py:206-215)"""
    al = ts.AlanineDipeptideVacuum()
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, n_steps=50, reassign_velocities=True, splitting='V R O R V')
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=6, engine=hip_engine_factory(), seed=4, online_analysis_interval=None)
    s.create(states.ThermodynamicState(al.system, 300.0), [states.SamplerState(al.positions)], storage=None,
             min_temperature=300.0, max_temperature=450.0, n_temperatures=4)
    s.run()
    u = s.energy_thermodynamic_states
    assert s.iteration == 6 and np.all(np.isfinite(u))
    T = np.array([t.temperature for t in s._thermodynamic_states])
    assert np.allclose(u * T[None, :], (u[:, :1] * T[0]), rtol=1e-12)
    assert np.abs(u[:, 0] * KB * T[0]).max() < 500.0


def _water_cluster(n=12):
    """n rigid TIP3P waters on a lattice without a box (testsystems.WaterCluster's shape: NoCutoff, three constraints per molecule)"""
    s = System()
    nb = NonbondedForce(); nb.setNonbondedMethod(NonbondedForce.NoCutoff)
    dOH, ang = 0.09572, np.deg2rad(104.52)
    dHH = 2.0 * dOH * np.sin(0.5 * ang)
    x = []
    g = np.stack(np.meshgrid(np.arange(3), np.arange(2), np.arange(2), indexing='ij'), axis=-1).reshape(-1, 3)[:n] * 0.31
    for m in range(n):
        o = 3 * m
        for mass, q, sg, ep in ((15.9994, -0.834, 0.3150752406575124, 0.635968), (1.008, 0.417, 1.0, 0.0), (1.008, 0.417, 1.0, 0.0)):
            s.addParticle(mass); nb.addParticle(q, sg, ep)
        s.addConstraint(o, o + 1, dOH); s.addConstraint(o, o + 2, dOH); s.addConstraint(o + 1, o + 2, dHH)
        nb.addException(o, o + 1, 0.0, 1.0, 0.0); nb.addException(o, o + 2, 0.0, 1.0, 0.0); nb.addException(o + 1, o + 2, 0.0, 1.0, 0.0)
        x += [g[m], g[m] + [dOH, 0.0, 0.0], g[m] + [dOH * np.cos(ang), dOH * np.sin(ang), 0.0]]
    s.addForce(nb)
    return s, np.array(x, dtype=np.float64)


@pytest.mark.gpu
@pytest.mark.parametrize('which,splitting,dt,n_steps', [('alanine', 'V R R O R R V', 0.002, 10), ('alanine', 'V R R O R R V', 0.002, 500),
                                                        ('alanine', 'O V R V O', 0.001, 100), ('droplet', 'V R O R V', 0.002, 100), ('water', 'V R O R V', 0.001, 5), ('water', 'V R R O R R V', 0.002, 10)])
def test_resident_small_molecule_kernel_follows_the_regular_launches(hip_engine_factory, monkeypatch, which, splitting, dt, n_steps):
    """NoCutoff systems of up to 64 atoms are propagated by ONE launch per move (integrate.hip resident_mol_kernel: a workgroup per
    replica, a constraint unit per thread, positions and fixed-point force accumulators in LDS) instead of three dependent launches per
    MD step.  Pair sums in the same fp32 order, every contribution converted to fixed point the same way, the same Philox streams and
    the same centre-of-mass sum: the trajectory follows the regular launches (REMD_RESIDENT=0) to fp32 rounding -- one ulp of a velocity
    per step at first, amplified by the dynamics afterwards (10 steps: 1e-7 nm; 500 steps of a molecule at 600 K: the same basin,
    the same energy within a few kT)."""
    # (droplet: 60 free atoms, no constraint units, no listed terms; water: twelve rigid molecules, the analytic three-site solve)
    system, x0 = _cluster(60) if which == 'droplet' else _water_cluster() if which == 'water' else _system(which)
    desc = system_to_desc(system)
    out = []
    for flag in ('1', '0'):
        monkeypatch.setenv('REMD_RESIDENT', flag)
        eng = hip_engine_factory()
        eng.set_system(desc)
        T = np.array([300.0, 450.0, 600.0])
        eng.set_states(1.0 / (KB * T))
        eng.set_integrator(splitting, dt, 1.0, n_steps, True, 1e-8)
        eng.seed(11)
        x = np.stack([x0 + 0.001 * r * np.random.default_rng(r).normal(size=x0.shape) for r in range(3)])
        eng.set_replicas(3, 0, x, None, np.zeros((3, 3)), np.array([2, 0, 1]))
        for it in range(2):
            assert not eng.propagate(it).any()
        xa, va = eng.get_replicas()[:2]
        out.append((xa.copy(), va.copy(), eng.compute_energies(want_potential=True)[1]))
    (xa, va, ua), (xb, vb, ub) = out
    assert np.abs(xa - x).max() > (0.2 if n_steps == 500 else 0.01 if n_steps > 10 else 0.002)              # it moved
    dx, dv = np.abs(xa - xb), np.abs(va - vb)
    if n_steps <= 10:
        assert dx.max() < 2e-6 and dv.max() < 2e-4, (dx.max(), dv.max())
        assert np.allclose(ua, ub, rtol=1e-5, atol=1e-3)
    elif n_steps <= 100:
        assert dx.max() < 2e-4 and dv.max() < 2e-2, (dx.max(), dv.max())
        assert np.allclose(ua, ub, rtol=1e-3, atol=0.05)
    else:
        assert np.median(dx) < 5e-3, np.median(dx)
        assert np.abs(ua - ub).max() < 40.0, (ua, ub)
    if which in ('alanine', 'water'):          # constrained distances at their lengths on both paths
        for i, j, d in system.constraints:
            for xx in (xa, xb):
                assert np.abs(np.linalg.norm(xx[:, i] - xx[:, j], axis=-1) - d).max() < 5e-6 * d + 1e-6


def test_alchemical_vacuum_system_through_the_store_adapter():
    from openmmtools_amd import system_xml
    al = ts.AlanineDipeptideVacuum()
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(
        al.system, alchemy.AlchemicalRegion(alchemical_atoms=range(6), softcore_beta=0.2, alchemical_torsions=True, name='cap'))
    xml = system_xml.to_xml(system)
    # the custom forces copy the NonbondedForce's method: NoCutoff = 0, no switch on the electrostatics (alchemy.py:1793-1796, 1818-1824)
    import xml.etree.ElementTree as ET
    cnb = [f for f in ET.fromstring(xml).find('Forces').findall('Force') if f.get('type') == 'CustomNonbondedForce']
    assert len(cnb) == 4 and all(f.get('method') == '0' for f in cnb)
    elec = [f for f in cnb if 'U_electrostatics' in f.get('energy')]
    assert elec[0].get('energy') == G['expressions']['electrostatics_nocutoff'].replace('lambda_electrostatics', 'lambda_electrostatics_cap') and elec[0].get('useSwitchingFunction') == '0'
    back, _ = system_xml.from_xml(xml)
    da, db = system_to_desc(system), system_to_desc(back)
    assert db['nb_method'] == 3 and back.alchemical_regions[0].name == 'cap'
    for k in da:
        if k != 'alch_regions':
            assert np.array_equal(np.asarray(da[k]), np.asarray(db[k])), k
    for k in da['alch_regions']:
        if not k.endswith('_index'):
            assert np.array_equal(np.asarray(da['alch_regions'][k]), np.asarray(db['alch_regions'][k])), k
