"""GPU: NPT states = Monte Carlo barostat inside the Langevin step (forces.hip: baro_* kernels, remd_set_barostat) and
the beta p V term of u_kl, against the f64 oracle restatement of OpenMM's MonteCarloBarostatImpl (the machinery the
reference's NPT ThermodynamicState relies on, states.py:1177-1181)."""
import numpy as np
import pytest
from openmmtools_amd import testsystems as ts, states, mcmc, unit
from openmmtools_amd.system import system_to_desc
from openmmtools_amd.multistate import ParallelTemperingSampler
from oracle import md_oracle as mo
from oracle.forcefield import ForceFieldOracle
from oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu
KB = 0.008314462618153242


def _setup(eng, system, x, T, pressure, n_steps, seed=11, dt=0.002):
    R = len(x)
    desc = system_to_desc(system)
    eng.set_system(desc)
    eng.set_states(1.0 / (KB * np.asarray(T, dtype=np.float64)))
    eng.set_integrator('V R O R V', dt, 1.0, n_steps, True, 1e-8)
    eng.set_barostat(np.full(R, pressure), 25)
    eng.seed(seed)
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (R, 1))
    eng.set_replicas(R, 0, x, None, box, np.arange(R))
    return desc, box


def test_barostat_tracks_the_oracle_on_the_lj_fluid(hip_engine_factory):
    lj = ts.LennardJonesFluid(nparticles=216)
    R = 3
    rng = np.random.default_rng(1)
    x = np.stack([lj.positions + 0.005 * rng.normal(size=lj.positions.shape) for _ in range(R)])
    T = [110.0, 120.0, 130.0]
    p = 40.0 * unit.bar
    eng, ora = hip_engine_factory(), OracleEngine(ForceFieldOracle)
    _setup(eng, lj.system, x, T, p, 50)
    _setup(ora, lj.system, x, T, p, 50)
    V0 = np.prod(eng.get_boxes(), axis=1)
    for it in range(2):                                        # 2 x 50 steps = 4 volume moves per replica
        assert not eng.propagate(it).any()
        ora.propagate(it)
        Vd, Vo = np.prod(eng.get_boxes(), axis=1), np.prod(ora.get_boxes(), axis=1)
        assert np.allclose(Vd, Vo, rtol=2e-5), (it, Vd, Vo)
    assert np.all(Vd != V0)
    vs, na, nc = eng.barostat_stats()
    assert na.tolist() == [4, 4, 4]
    assert nc.tolist() == [ora._baro.state[r][4] for r in range(R)]
    # u_kl rows include beta_l p_l V_r; energies evaluated by the oracle on the device's positions and boxes
    xg, _, _, _ = eng.get_replicas()
    rows = eng.compute_energies()
    sysm = ForceFieldOracle(system_to_desc(lj.system))
    beta = 1.0 / (KB * np.array(T))
    boxes = eng.get_boxes()
    for r in range(R):
        U = sysm.potential(xg[r], boxes[r])
        expect = beta * (U + p * np.prod(boxes[r]))
        assert np.allclose(rows[r], expect, rtol=1e-5, atol=1e-4)


def test_ideal_gas_volume_distribution_on_device(hip_engine_factory):
    """eps = 0, q = 0: P(V) ~ V^N exp(-beta p V)  =>  <V> = (N + 1) kT / p  (and the same for the oracle, CPU test)."""
    N, T, p = 64, 300.0, 30.0 * unit.bar
    lj = ts.LennardJonesFluid(nparticles=N, epsilon=0.0)
    R = 8
    x = np.tile(lj.positions, (R, 1, 1))
    eng = hip_engine_factory()
    _setup(eng, lj.system, x, [T] * R, p, 25, dt=0.001)
    vols = []
    for it in range(700):
        assert not eng.propagate(it).any()
        if it >= 200:
            vols.append(np.prod(eng.get_boxes(), axis=1))
    vols = np.array(vols)
    expect = (N + 1) * KB * T / p
    mean = vols.mean()
    sem = vols.std() / np.sqrt(vols.size / 15.0)
    assert abs(mean - expect) < 5 * sem + 0.01 * expect, (mean, expect, sem)
    rel = vols.std() / mean
    assert 0.8 / np.sqrt(N + 1) < rel < 1.25 / np.sqrt(N + 1)          # Gamma(N + 1) distribution: sigma / mean = (N + 1)^-1/2
    vs, na, nc = eng.barostat_stats()
    assert np.all(na == 700) and np.all((nc / na > 0.2) & (nc / na < 0.9))      # adaptation aims at 25-75 % per 10-move window


def test_alanine_npt_keeps_molecules_rigid(hip_engine_factory):
    al = ts.AlanineDipeptideExplicit()
    R = 2
    x = np.stack([al.positions, al.positions])
    eng = hip_engine_factory()
    desc, box0 = _setup(eng, al.system, x, [300.0, 310.0], 1.0 * unit.bar, 100)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 100, True, 1e-8)
    assert not eng.propagate(0).any()
    boxes = eng.get_boxes()
    assert np.all(np.abs(boxes / box0 - 1.0) < 0.02) and np.any(boxes != box0)
    xg, _, ug, _ = eng.get_replicas(potential=True)
    cons = mo.OracleSystem(desc).constraints
    for (i, j, d0) in cons[:600]:
        assert abs(np.linalg.norm(xg[0][i] - xg[0][j]) - d0) < 3e-6
    sysm = ForceFieldOracle(desc)
    U = sysm.potential(xg[1], boxes[1])
    assert np.isclose(ug[1], U, rtol=1e-5), (ug[1], U)
    vs, na, nc = eng.barostat_stats()
    assert na.tolist() == [4, 4]


def test_sampler_npt_on_device(hip_engine_factory):
    lj = ts.LennardJonesFluid(nparticles=216)
    tstate = states.ThermodynamicState(lj.system, 120.0, pressure=40.0 * unit.bar)
    ss = states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                              n_steps=50, reassign_velocities=True, splitting='V R O R V')
    res = []
    for engine in (hip_engine_factory(), OracleEngine(ForceFieldOracle)):
        s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=2, engine=engine, seed=9)
        s.create(tstate, [ss], min_temperature=110.0, max_temperature=130.0, n_temperatures=3)
        s.run()
        s._sampler_states_stale = True
        s._sync_sampler_states()
        res.append((s.replica_thermodynamic_states.copy(), np.array([st.volume for st in s.sampler_states]),
                    s.energy_thermodynamic_states.copy()))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.allclose(res[0][1], res[1][1], rtol=5e-5)
    assert np.allclose(res[0][2], res[1][2], rtol=2e-4, atol=2e-3)


def test_npt_with_alchemical_states_tracks_the_oracle(hip_engine_factory):
    """NPT + lambda_sterics states (production free-energy mode): the per-state long-range constants scale as 1/V with each
    replica's box, both in u_kl and in the barostat's acceptance."""
    from openmmtools_amd import alchemy
    from openmmtools_amd.system import NonbondedForce
    lj = ts.LennardJonesFluid(nparticles=216)
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(10))
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, region)
    nb = [f for f in system.getForces() if isinstance(f, NonbondedForce)][0]
    lam = np.array([1.0, 0.6, 0.3, 0.0])
    R = len(lam)
    V0 = float(np.prod(np.diag(system.getDefaultPeriodicBoxVectors())))
    econst = alchemy.alchemical_long_range_constants(system, nb, lam, V0)
    assert np.any(econst != 0.0)
    p = 40.0 * unit.bar
    x = np.tile(lj.positions, (R, 1, 1))
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (R, 1))
    engines = []
    for eng in (hip_engine_factory(), OracleEngine(ForceFieldOracle)):
        eng.set_system(system_to_desc(system))
        eng.set_states(np.full(R, 1.0 / (KB * 120.0)), lam, None, econst)
        eng.set_integrator('V R O R V', 0.002, 1.0, 50, True, 1e-8)
        eng.set_barostat(np.full(R, p), 25)
        eng.set_energy_const_volume(V0)
        eng.seed(21)
        eng.set_replicas(R, 0, x, None, box, np.arange(R))
        engines.append(eng)
    dev, ora = engines
    for it in range(2):
        assert not dev.propagate(it).any()
        ora.propagate(it)
    Vd, Vo = np.prod(dev.get_boxes(), axis=1), np.prod(ora.get_boxes(), axis=1)
    assert np.allclose(Vd, Vo, rtol=3e-5), (Vd, Vo)
    assert np.all(Vd != V0)
    # u_kl of the device against the oracle evaluated on the device's own configuration
    xg, _, _, _ = dev.get_replicas()
    ora.x, ora.box = xg.copy(), dev.get_boxes()
    assert np.allclose(dev.compute_energies(), ora.compute_energies(), rtol=1e-5, atol=2e-4)


def test_barostat_acceptance_sees_the_lambda_controlled_pairs(hip_engine_factory):
    """The volume move's Metropolis test differences the potential of the replica's OWN state (MonteCarloBarostat on the
    alchemical System's Context in the reference).  Until round 4 the device's per-replica potential left the
    lambda_sterics-controlled pairs out (they only entered the u_kl rows), so the barostat of an alchemical NPT run did not
    feel the guest / solvent Lennard-Jones energy.  Dense start (close alchemical / solvent contacts: that energy changes by
    many kT under a volume move): 16 attempts per replica decide as the f64 oracle does, and the potential is the oracle's
    full energy at each replica's lambda."""
    from openmmtools_amd import alchemy
    lj = ts.LennardJonesFluid(nparticles=216, reduced_density=0.7)
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, alchemy.AlchemicalRegion(alchemical_atoms=range(10)))
    lam = np.array([1.0, 0.8, 0.4, 0.0])
    R = len(lam)
    x = np.tile(lj.positions, (R, 1, 1))
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (R, 1))
    desc = system_to_desc(system)
    engines = []
    for eng in (hip_engine_factory(), OracleEngine(ForceFieldOracle)):
        eng.set_system(desc)
        eng.set_states(np.full(R, 1.0 / (KB * 120.0)), lam, None, None)
        eng.set_integrator('V R O R V', 0.001, 1.0, 5, True, 1e-8)
        eng.set_barostat(np.full(R, 40.0 * unit.bar), 25)
        eng.seed(4)
        eng.set_replicas(R, 0, x, None, box, np.arange(R))
        engines.append(eng)
    dev, ora = engines
    ff = ForceFieldOracle(desc)
    U = dev.compute_energies(want_potential=True)[1]
    full = np.array([ff.energy_forces(x[r], box[r], lambda_sterics=lam[r], forces=False)[0] for r in range(R)])
    assert np.allclose(U, full, rtol=1e-5), (U, full)
    assert abs(full[0] - full[-1]) > 50.0                 # the lambda-controlled pairs matter here
    assert np.allclose(dev.get_replicas(potential=True)[2], full, rtol=1e-5)
    volumes = []
    for _ in range(4):
        dev.barostat_attempts(4)
        ora.barostat_attempts(4)
        volumes.append((np.prod(dev.get_boxes(), axis=1), np.prod(ora.get_boxes(), axis=1)))
    for vd, vo in volumes:
        assert np.allclose(vd, vo, rtol=3e-5), volumes
    assert len({tuple(np.round(vd, 9)) for vd, _ in volumes}) > 1          # moves were accepted along the way


def test_hostguest_npt_alchemical_production_mode(hip_engine_factory):
    """CB7:B2 host-guest, lambda_electrostatics + lambda_sterics states, NPT at 1 bar (the production free-energy ensemble):
    50 g-BAOAB steps with two volume moves per replica on the device, then u_kl = beta_l (U(l) + c_l V0/V + p V) against
    the oracle evaluated on the device's configuration and boxes."""
    from openmmtools_amd import alchemy
    from openmmtools_amd.system import NonbondedForce
    hg = ts.HostGuestExplicit()
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(126, 156))
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(hg.system, region)
    lam_e = np.array([1.0, 0.5, 0.0, 0.0])
    lam_s = np.array([1.0, 1.0, 1.0, 0.4])
    K = len(lam_e)
    nb = [f for f in system.getForces() if isinstance(f, NonbondedForce)][0]
    box0 = np.diag(system.getDefaultPeriodicBoxVectors())
    V0 = float(np.prod(box0))
    econst = alchemy.alchemical_long_range_constants(system, nb, lam_s, V0)
    p = 1.0 * unit.bar
    eng = hip_engine_factory()
    desc = system_to_desc(system)
    eng.set_system(desc)
    beta = 1.0 / (KB * 300.0)
    eng.set_states(np.full(K, beta), lam_s, lam_e, econst)
    eng.set_integrator('V R R O R R V', 0.002, 1.0, 50, True, 1e-8)
    eng.set_barostat(np.full(K, p), 25)
    eng.set_energy_const_volume(V0)
    eng.seed(5)
    labels = np.array([1, 3])
    x = np.stack([hg.positions, hg.positions])
    eng.set_replicas(2, 0, x, None, np.tile(box0, (2, 1)), labels)
    assert not eng.propagate(0).any()
    boxes = eng.get_boxes()
    assert np.any(boxes != box0) and np.all(np.abs(boxes / box0 - 1.0) < 0.02)
    _, na, _ = eng.barostat_stats()
    assert na.tolist() == [2, 2]
    rows = eng.compute_energies()
    xd = eng.get_replicas()[0]
    ff = ForceFieldOracle(desc)
    for r in range(2):
        V = float(np.prod(boxes[r]))
        ref = beta * (ff.state_energies(xd[r], boxes[r], lam_s, lam_e) + econst * V0 / V + p * V)
        assert np.allclose(rows[r], ref, rtol=1e-5), np.abs(rows[r] / ref - 1).max()


def test_barostat_attempts_outside_the_integrator_track_the_oracle(hip_engine_factory):
    """remd_barostat_attempts (MonteCarloBarostatMove, mcmc.py:1597-1700): explicit volume moves share the move, the random
    stream and the attempt counter with the in-integrator barostat."""
    lj = ts.LennardJonesFluid(nparticles=216)
    R = 2
    rng = np.random.default_rng(5)
    x = np.stack([lj.positions + 0.005 * rng.normal(size=lj.positions.shape) for _ in range(R)])
    T = [110.0, 125.0]
    p = 40.0 * unit.bar
    eng, ora = hip_engine_factory(), OracleEngine(ForceFieldOracle)
    _setup(eng, lj.system, x, T, p, 25)
    _setup(ora, lj.system, x, T, p, 25)
    V0 = np.prod(eng.get_boxes(), axis=1)
    eng.barostat_attempts(3); ora.barostat_attempts(3)
    Vd, Vo = np.prod(eng.get_boxes(), axis=1), np.prod(ora.get_boxes(), axis=1)
    assert np.allclose(Vd, Vo, rtol=2e-5) and np.all(Vd != V0)
    assert np.abs(eng.get_replicas()[0] - ora.x).max() < 2e-5
    assert not eng.propagate(0).any()                          # 25 steps: in-integrator attempt number 3 follows
    ora.propagate(0)
    assert np.allclose(np.prod(eng.get_boxes(), axis=1), np.prod(ora.get_boxes(), axis=1), rtol=2e-5)
    assert eng.barostat_stats()[1].tolist() == [4, 4]
    nvt = hip_engine_factory()
    desc = system_to_desc(lj.system)
    nvt.set_system(desc); nvt.set_states(1.0 / (KB * np.array(T)))
    nvt.set_replicas(R, 0, x, None, np.tile(np.diag(lj.system.getDefaultPeriodicBoxVectors()), (R, 1)), np.arange(R))
    with pytest.raises(RuntimeError):
        nvt.barostat_attempts(1)


def test_a_box_below_twice_the_cutoff_is_an_error_not_a_broken_minimum_image(hip_engine_factory):
    """an ideal gas under a pressure whose equilibrium volume is (1.5 nm)^3 shrinks towards a box of 1.5 nm, the cutoff is
    1.02 nm.  The Monte Carlo barostat must never ACCEPT a trial box with an edge below twice the cutoff (its energy was evaluated with a
    broken minimum image) and the propagation that proposed it fails the way OpenMM's does ("... less than twice the nonbonded cutoff");
    every box the handle ever held stays legal."""
    N, T = 64, 300.0
    lj = ts.LennardJonesFluid(nparticles=N, epsilon=0.0)
    cutoff = 1.02
    p = (N + 1) * KB * T / 1.5 ** 3
    R = 4
    eng = hip_engine_factory()
    _setup(eng, lj.system, np.tile(lj.positions, (R, 1, 1)), [T] * R, p, 25, dt=0.001)
    assert np.all(eng.get_boxes() >= 2.0 * cutoff)
    with pytest.raises(RuntimeError, match='twice the nonbonded cutoff'):
        for it in range(3000):
            eng.propagate(it)
            assert np.all(eng.get_boxes() >= 2.0 * cutoff)
    assert np.all(eng.get_boxes() >= 2.0 * cutoff)
