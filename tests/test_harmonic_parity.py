"""GPU parity on BASELINE config 1 (testsystems.HarmonicOscillator, parallel tempering, BAOAB):
integrator substeps, Maxwell-Boltzmann draw, u_kl assembly and the whole mix->propagate->u_kl
iteration against the f64 oracle on the same Philox stream.

Tolerances: the device state is fp32, the oracle f64.  Single substeps agree to a few fp32 ulps of
the quantities involved; u_kl to 1e-5 relative (north_star), checked on positions copied from the
device so that only the energy arithmetic is compared."""
import numpy as np
import pytest
from openmmtools_amd import testsystems, states, mcmc, unit
from openmmtools_amd.system import system_to_desc
from openmmtools_amd.multistate import ParallelTemperingSampler
from oracle import md_oracle as mo
from oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu
SEED = 0xC0FFEE
KB = 0.008314462618153242


def _setup(engine, R=4, n_steps=50, splitting='V R O R V', dt=0.001, x=None, v=None):
    ho = testsystems.HarmonicOscillator()
    desc = system_to_desc(ho.system)
    engine.set_system(desc)
    T = np.logspace(np.log10(300.0), np.log10(600.0), R)
    engine.set_states(1.0 / (KB * T))
    engine.set_integrator(splitting, dt, 1.0, n_steps, True, 1e-8)
    engine.seed(SEED)
    rng = np.random.default_rng(0)
    x = rng.normal(scale=0.01, size=(R, 1, 3)) if x is None else x
    v = rng.normal(scale=0.3, size=(R, 1, 3)) if v is None else v
    engine.set_replicas(R, 0, x, v, np.zeros((R, 3)), np.arange(R))
    return desc, T, x, v


def test_substeps_match_oracle(hip_engine_factory):
    eng, ora = hip_engine_factory(), OracleEngine()
    _, T, x, v = _setup(eng)
    _setup(ora)
    for tok, step in (('V', 0), ('R', 0), ('O', 0), ('O', 3), ('R', 1), ('V', 1)):
        eng.step(tok, iteration=2, first_step=step)
        ora.step(tok, iteration=2, first_step=step)
        xg, vg, _, _ = eng.get_replicas()
        assert np.allclose(xg, ora.x, rtol=2e-6, atol=1e-8), tok
        assert np.allclose(vg, ora.v, rtol=2e-5, atol=2e-6), tok
        # re-synchronise the oracle on the device's fp32 state so that errors do not accumulate
        ora.x, ora.v = xg.copy(), vg.copy()


def test_forces_and_energy(hip_engine_factory):
    eng = hip_engine_factory()
    desc, T, x, v = _setup(eng)
    f = eng.get_forces()
    x32 = x.astype(np.float32).astype(np.float64)
    assert np.allclose(f, -desc['ext_K'] * x32, rtol=1e-6)
    rows, U = eng.compute_energies(want_potential=True)
    Uref = 0.5 * desc['ext_K'] * (x32 ** 2).sum(axis=(1, 2))
    assert np.allclose(U, Uref, rtol=1e-6)
    beta = 1.0 / (KB * T)
    assert np.allclose(rows, U[:, None] * beta[None, :], rtol=1e-14)         # paralleltempering.py:206-215
    assert np.allclose(rows, Uref[:, None] * beta[None, :], rtol=1e-5)       # north_star tolerance
    _, _, _, ke = eng.get_replicas(kinetic=True)
    v32 = v.astype(np.float32).astype(np.float64)
    assert np.allclose(ke, 0.5 * 39.948 * (v32 ** 2).sum(axis=(1, 2)), rtol=1e-6)


def test_maxwell_boltzmann_matches_oracle_stream(hip_engine_factory):
    eng, ora = hip_engine_factory(), OracleEngine()
    _setup(eng, n_steps=0)
    _setup(ora, n_steps=0)
    eng.propagate(5)
    ora.propagate(5)
    _, vg, _, _ = eng.get_replicas()
    assert np.allclose(vg, ora.v, rtol=3e-5, atol=1e-6)


@pytest.mark.parametrize('splitting', ['V R O R V', 'V R R O R R V', 'O V R V O'])
def test_short_trajectory_tracks_oracle(hip_engine_factory, splitting):
    """20 steps of a stable harmonic system: fp32 vs f64 stays within 1e-4 relative of the amplitude."""
    eng, ora = hip_engine_factory(), OracleEngine()
    _setup(eng, n_steps=20, splitting=splitting)
    _setup(ora, n_steps=20, splitting=splitting)
    assert not eng.propagate(1).any()
    ora.propagate(1)
    xg, vg, _, _ = eng.get_replicas()
    assert np.allclose(xg, ora.x, rtol=0, atol=1e-4 * np.abs(ora.x).max())
    assert np.allclose(vg, ora.v, rtol=0, atol=1e-4 * np.abs(ora.v).max())


def test_full_iteration_loop_against_oracle_sampler(hip_engine_factory):
    """ParallelTemperingSampler.run on the device vs the same sampler on the oracle engine: u_kl within 1e-5
    relative each iteration is NOT expected (chaotic fp32 drift is tiny here but nonzero), so the check is:
    device u_kl -> oracle mixing == device mixing (bit exact), and energies stay 1e-3-close over 5 iterations."""
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0)
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                              n_steps=100, reassign_velocities=True, splitting='V R O R V')
    dev = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=5, engine=hip_engine_factory(), seed=SEED)
    ora = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=5, engine=OracleEngine(), seed=SEED)
    for s in (dev, ora):
        s.create(ts, [ss], min_temperature=300.0, max_temperature=600.0, n_temperatures=4)
    import oracle
    for it in range(1, 6):
        u_before = dev.energy_thermodynamic_states.copy() if it > 1 else None
        labels_before = dev.replica_thermodynamic_states.copy()
        dev.run(1)
        ora.run(1)
        if u_before is not None:
            ref = oracle.mix('swap-all', SEED, it, u_before, labels_before)
            assert np.array_equal(ref[0], dev.replica_thermodynamic_states)
            assert np.array_equal(ref[1], dev._n_accepted_matrix) and np.array_equal(ref[2], dev._n_proposed_matrix)
        assert np.allclose(dev.energy_thermodynamic_states, ora.energy_thermodynamic_states, rtol=2e-3, atol=1e-4)
    assert np.array_equal(dev.replica_thermodynamic_states, ora.replica_thermodynamic_states)


def test_equipartition_statistics(hip_engine_factory):
    """tests/test_mcmc.py:178-203 / testsystems.py:804-840: <U> = 3/2 kT for the oscillator under Langevin."""
    eng = hip_engine_factory()
    R = 64
    ho = testsystems.HarmonicOscillator()
    eng.set_system(system_to_desc(ho.system))
    T = np.full(R, 300.0)
    eng.set_states(1.0 / (KB * T))
    eng.set_integrator('V R O R V', 0.002, 20.0, 200, False, 1e-8)
    eng.seed(99)
    eng.set_replicas(R, 0, np.zeros((R, 1, 3)), np.zeros((R, 1, 3)), np.zeros((R, 3)), np.arange(R))
    samples = []
    for it in range(60):
        eng.propagate(it)
        if it >= 10:
            samples.append(eng.compute_energies(want_potential=True)[1])
    U = np.concatenate(samples)
    expect = 1.5 * KB * 300.0
    sem = U.std() / np.sqrt(len(U) / 2.0)
    assert abs(U.mean() - expect) < 6.0 * sem, (U.mean(), expect, sem)


def test_nan_restart_bookkeeping(hip_engine_factory):
    """remd_set_restart_attempts (mcmc.py:706-759): a replica that fails every attempt is flagged, and the replicas
    that were fine keep exactly the result of their first (successful) attempt although the batch was re-run."""
    clean, eng = hip_engine_factory(), hip_engine_factory()
    R = 4
    rng = np.random.default_rng(1)
    x = rng.normal(scale=0.01, size=(R, 1, 3))
    _setup(clean, R=R, x=x)
    flags = clean.propagate(5)
    assert not flags.any()
    xc, vc, _, _ = clean.get_replicas()
    xbad = x.copy()
    xbad[2] = np.nan
    _setup(eng, R=R, x=xbad)
    eng.set_restart_attempts(3)
    flags = eng.propagate(5)
    assert flags.tolist() == [0, 0, 1, 0]
    xg, vg, _, _ = eng.get_replicas()
    for r in (0, 1, 3):
        assert np.array_equal(xg[r], xc[r]) and np.array_equal(vg[r], vc[r])
    assert not np.isfinite(xg[2]).all()
    # the state stays usable: energies of the healthy replicas are finite, the next propagate flags the same replica
    u = eng.compute_energies()
    assert np.isfinite(u[[0, 1, 3]]).all()
    assert eng.propagate(6).tolist() == [0, 0, 1, 0]


def test_storage_and_resume_on_device(hip_engine_factory, tmp_path):
    """SURVEY 8(f) row 1 on the device engine: every iteration's u_kl / labels / statistics reach storage, checkpoints
    hold the f4 snapshot read back from the GPU, and from_storage continues; the first mix after the resume uses the
    stored energies (multistatesampler.py:1003-1020), checked against the sequential oracle on that matrix."""
    import oracle
    from openmmtools_amd.multistate import MultiStateReporter
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0)
    ss = states.SamplerState(ho.positions)
    move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                              n_steps=20, reassign_velocities=True, splitting='V R O R V')
    path = str(tmp_path / 'run.nc')
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=7, engine=hip_engine_factory(), seed=SEED)
    s.create(ts, [ss], storage=MultiStateReporter(path, checkpoint_interval=2), min_temperature=300.0,
             max_temperature=600.0, n_temperatures=6)
    for it in range(1, 5):
        s.run(1)
        e = MultiStateReporter(path, open_mode='r').read_energies(it)[0]
        assert np.array_equal(e, s.energy_thermodynamic_states)
    x4 = np.stack([st.positions for st in s.sampler_states])
    rd = MultiStateReporter(path, open_mode='r')
    assert rd.read_checkpoint_iterations() == [0, 2, 4]
    cp = np.stack([c.positions for c in rd.read_sampler_states(4)])
    assert np.array_equal(cp, x4.astype(np.float32).astype(np.float64))
    e4, lab4 = rd.read_energies(4)[0], rd.read_replica_thermodynamic_states(4)
    r = ParallelTemperingSampler.from_storage(path, engine=hip_engine_factory())
    assert r.iteration == 4
    r.run(1)
    ref = oracle.mix('swap-all', SEED, 5, e4, lab4)
    assert np.array_equal(r.replica_thermodynamic_states, ref[0])
    assert np.array_equal(r._n_accepted_matrix, ref[1]) and np.array_equal(r._n_proposed_matrix, ref[2])
    r.run()
    assert r.iteration == 7 and np.isfinite(r.energy_thermodynamic_states).all()
    assert MultiStateReporter(path, open_mode='r').read_energies()[0].shape == (8, 6, 6)
