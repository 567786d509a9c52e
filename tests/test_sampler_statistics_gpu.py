"""GPU: statistical known answers at the sampler level, mirroring the reference's own whole-sampler tests
(openmmtools/tests/test_sampling.py:93-451: harmonic oscillators, analytic free energies, MBAR within 6 sigma).

Our engine's states differ in temperature (and lambda), so the analytic ladder is the 3-D harmonic oscillator at
temperatures T_k:  f_k = -ln Z_k,  Z_k = (2 pi kT_k / K)^{3/2}  =>  f_k - f_0 = -(3/2) ln(T_k / T_0)."""
import numpy as np
import pytest
from openmmtools_amd import testsystems, states, mcmc, unit
from openmmtools_amd.multistate import ParallelTemperingSampler, SAMSSampler

pytestmark = pytest.mark.gpu


def _mbar(u_kn, N_k, tol=1e-10, max_iter=5000):
    """Self-consistent MBAR (Shirts & Chodera 2008, eq. 11) for a small problem; returns f_k with f_0 = 0."""
    K, N = u_kn.shape
    f = np.zeros(K)
    logN = np.log(N_k)
    for _ in range(max_iter):
        a = logN[:, None] + f[:, None] - u_kn                       # [K, N]
        m = a.max(axis=0)
        log_denom = m + np.log(np.exp(a - m).sum(axis=0))           # [N]
        b = -u_kn - log_denom[None, :]
        mb = b.max(axis=1)
        f_new = -(mb + np.log(np.exp(b - mb[:, None]).sum(axis=1)))
        f_new -= f_new[0]
        if np.abs(f_new - f).max() < tol:
            f = f_new
            break
        f = f_new
    return f


def _oscillator(engine, cls, n_iter, **kw):
    ho = testsystems.HarmonicOscillator()
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=20.0 / unit.picosecond,
                                              n_steps=250, reassign_velocities=True, splitting='V R O R V')
    return ho, ss, move


def test_parallel_tempering_free_energies_match_analytic(hip_engine_factory):
    ho, ss, move = _oscillator(None, None, None)
    ts = states.ThermodynamicState(ho.system, 300.0)
    n_iter, K = 400, 5
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=n_iter, engine=hip_engine_factory(), seed=2024)
    s.create(ts, [ss], min_temperature=300.0, max_temperature=600.0, n_temperatures=K)
    s.equilibrate(20)
    u, lab, acc = [], [], 0
    for _ in range(n_iter):
        s.run(1)
        u.append(s.energy_thermodynamic_states.copy())
        lab.append(s.replica_thermodynamic_states.copy())
        acc += s._n_accepted_matrix.sum()
    u = np.array(u)                               # [iter, R, K]
    lab = np.array(lab)
    assert acc > 0 and len(set(lab[:, 0])) == K    # replica 0 visits every temperature
    # decorrelated enough: reassign_velocities + 0.5 ps per iteration on a 0.1 ps oscillator
    u_kn = u.reshape(-1, K).T
    N_k = np.array([(lab == k).sum() for k in range(K)], dtype=float)
    f = _mbar(u_kn, N_k)
    T = np.array([st.temperature for st in s.thermodynamic_states])
    f_exact = -1.5 * np.log(T / T[0])
    # 6-sigma bar with a conservative per-state error estimate from block averaging (4 blocks)
    blocks = np.array([_mbar(b.reshape(-1, K).T, np.array([(lb == k).sum() for k in range(K)], dtype=float))
                       for b, lb in zip(np.array_split(u, 4), np.array_split(lab, 4))])
    sigma = blocks.std(axis=0, ddof=1) / np.sqrt(4) + 1e-3
    assert np.all(np.abs(f - f_exact) < 6.0 * sigma + 0.02), (f, f_exact, sigma)
    # equipartition at every temperature: <U> = 3/2 kT (testsystems.py:804-840)
    for k in range(K):
        Uk = np.concatenate([u[it, lab[it] == k, k] for it in range(n_iter)]) / s.thermodynamic_states[k].beta
        sem = Uk.std() / np.sqrt(len(Uk))
        assert abs(Uk.mean() - 1.5 * 0.008314462618153242 * T[k]) < 6 * sem


def test_sams_sampler_runs_on_device_and_flattens(hip_engine_factory):
    """SAMSSampler global-jump on the device: histogram bookkeeping (tests/test_sampling.py:2757-2787) and logZ
    estimates heading towards the analytic values."""
    ho, ss, move = _oscillator(None, None, None)
    T = np.linspace(300.0, 420.0, 4)
    sts = [states.ThermodynamicState(ho.system, t) for t in T]
    s = SAMSSampler(mcmc_moves=move, number_of_iterations=600, engine=hip_engine_factory(), seed=7, gamma0=1.0,
                    flatness_criteria='minimum-visits')
    s.create(sts, [ss, ss], storage=None)
    seen = [int(x) for x in s.replica_thermodynamic_states]     # create() reported iteration 0 (sams.py:381-393 counts it)
    orig = s._report_iteration

    def rec():
        orig()
        seen.extend(int(x) for x in s.replica_thermodynamic_states)
    s._report_iteration = rec
    s.run()
    assert np.array_equal(s._state_histogram, np.bincount(seen, minlength=4))
    assert s._stage == 1                                             # every state visited -> asymptotic stage
    assert s._state_histogram.min() > 0.1 * s._state_histogram.sum() / 4
    f_exact = -1.5 * np.log(T / T[0])
    assert np.abs((s._logZ - s._logZ[0]) - (-(f_exact - f_exact[0]))).max() < 0.5     # logZ = -f, loose SAMS bar


def test_unsampled_states_on_device(hip_engine_factory):
    """Unsampled end states ride along in the device u_kl rows (leading dimension K + U); only the K sampled columns are
    mixed (multistatesampler.py:1436-1456)."""
    from openmmtools_amd import alchemy
    from openmmtools_amd.multistate import ReplicaExchangeSampler
    from openmmtools_amd.system import system_to_desc
    from oracle.forcefield import ForceFieldOracle
    lj = testsystems.LennardJonesFluid(nparticles=216)
    region = alchemy.AlchemicalRegion(alchemical_atoms=range(10))
    system = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, region)

    def st(l):
        return states.CompoundThermodynamicState(states.ThermodynamicState(system, 120.0), [states.AlchemicalState(lambda_sterics=l)])
    sampled, unsampled = [st(l) for l in (0.8, 0.6, 0.4, 0.2)], [st(1.0), st(0.0)]
    ss = states.SamplerState(lj.positions, box_vectors=system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                              n_steps=20, reassign_velocities=True, splitting='V R O R V')
    s = ReplicaExchangeSampler(mcmc_moves=move, number_of_iterations=2, engine=hip_engine_factory(), seed=6)
    s.create(sampled, [ss], unsampled_thermodynamic_states=unsampled)
    s.run()
    assert s._energy_unsampled_states.shape == (4, 2) and s._n_proposed_matrix.sum() == 2 * 4 ** 3
    s._sampler_states_stale = True
    s._sync_sampler_states()
    ff = ForceFieldOracle(system_to_desc(system))
    lam = np.array([0.8, 0.6, 0.4, 0.2, 1.0, 0.0])
    econst = s._state_energy_constants(sampled + unsampled)
    beta = sampled[0].beta
    box = np.diag(system.getDefaultPeriodicBoxVectors())
    for r in range(4):
        ref = beta * (ff.state_energies(s.sampler_states[r].positions, box, lam, np.ones(6)) + econst)
        got = np.concatenate([s.energy_thermodynamic_states[r], s._energy_unsampled_states[r]])
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-4)
