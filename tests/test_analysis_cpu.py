"""Online / offline free-energy analysis: the numpy restatement of the pymbar algorithms the
reference's analyzer uses (openmmtools_amd/multistate/analysis.py), pinned against analytical results, and the
sampler plumbing of multistatesampler.py:1519-1735 (tests/test_sampling.py:100-300, 2213-2325 are the models)."""
import os
import numpy as np
import pytest
from openmmtools_amd import testsystems, states, mcmc, unit
from openmmtools_amd.multistate import ParallelTemperingSampler, MultiStateReporter
from openmmtools_amd.multistate import analysis as an
from oracle_engine import OracleEngine


def test_statistical_inefficiency_of_an_ar1_process():
    """g = (1 + rho) / (1 - rho) for x_t = rho x_{t-1} + noise."""
    rng = np.random.default_rng(1)
    for rho in (0.0, 0.5, 0.8):
        x = np.zeros(40000)
        e = rng.normal(size=x.size)
        for t in range(1, x.size):
            x[t] = rho * x[t - 1] + e[t]
        g = an.statistical_inefficiency(x)
        assert abs(g - (1 + rho) / (1 - rho)) < 0.15 * (1 + rho) / (1 - rho), (rho, g)
        assert an.statistical_inefficiency(x, fast=True) >= 1.0
    with pytest.raises(ValueError):
        an.statistical_inefficiency(np.ones(10))


def test_subsampling_and_equilibration_detection():
    assert an.subsample_correlated_data(np.zeros(12), g=2.5) == [0, 2, 5, 8, 10]          # int(round(n g)), half to even
    assert an.subsample_correlated_data(np.zeros(7), g=1.0) == list(range(7))
    assert an.subsample_correlated_data(np.zeros(7), g=2.2, conservative=True) == [0, 3, 6]
    # a relaxation followed by white noise: the detected origin sits after the transient
    rng = np.random.default_rng(2)
    u = rng.normal(size=600) + 30.0 * np.exp(-np.arange(600) / 15.0)
    i_t, g_i, n_eff = an.get_equilibration_data_per_sample(u, max_subset=100)
    assert len(i_t) == len(g_i) == len(n_eff) == 99 and i_t[0] == 6            # multistate/utils.py:190: the origin t = 0 is dropped
    t0 = i_t[n_eff.argmax()]
    assert 30 <= t0 <= 200
    # constant series: the special trap of multistate/utils.py:170-174
    i_t, g_i, n_eff = an.get_equilibration_data_per_sample(np.ones(5))
    assert list(g_i) == [1, 1, 1, 1] and list(n_eff) == [5, 4, 3, 2]


def test_time_series_helpers_of_multistate_utils():
    """multistate/utils.py:60-300, importable from both places the reference exposes them."""
    from openmmtools_amd import multistate
    from openmmtools_amd.multistate import utils as mu
    assert mu.get_equilibration_data is an.get_equilibration_data and multistate.remove_unequilibrated_data is an.remove_unequilibrated_data
    rng = np.random.default_rng(2)
    u = rng.normal(size=600) + 30.0 * np.exp(-np.arange(600) / 15.0)
    n_eq, g_t, n_eff = an.get_equilibration_data(u, max_subset=100)
    i_t, g_i, n_i = an.get_equilibration_data_per_sample(u, max_subset=100)
    assert (n_eq, g_t, n_eff) == (i_t[n_i.argmax()], g_i[n_i.argmax()], n_i.max())
    assert an.get_decorrelation_time(u[200:]) == an.statistical_inefficiency(u[200:])
    data = np.arange(24).reshape(2, 12)
    assert an.remove_unequilibrated_data(data, 5, axis=1).shape == (2, 7) and an.remove_unequilibrated_data(data, 1, axis=0).shape == (1, 12)
    assert np.array_equal(an.subsample_data_along_axis(data, 2.5, axis=1), data[:, [0, 2, 5, 8, 10]])
    assert an.generate_phase_name(None, ['phase0', 'phase1']) == 'phase2'
    assert an.generate_phase_name('complex', ['complex', 'complex0']) == 'complex1' and an.generate_phase_name('solvent', ['complex']) == 'solvent'
    assert issubclass(multistate.ParallelTemperingAnalyzer, multistate.ReplicaExchangeAnalyzer)
    assert issubclass(multistate.SAMSAnalyzer, multistate.MultiStateSamplerAnalyzer)


def _harmonic_samples(sigmas, sampled, n_per_state, rng):
    """Exact samples of 3-D harmonic oscillators u_k(x) = |x|^2 / (2 sigma_k^2); f_k = -3/2 ln(2 pi sigma_k^2)."""
    xs, N_k = [], np.zeros(len(sigmas), dtype=int)
    for k in sampled:
        xs.append(rng.normal(scale=sigmas[k], size=(n_per_state, 3)))
        N_k[k] = n_per_state
    x = np.concatenate(xs)
    r2 = (x ** 2).sum(axis=1)
    u_kn = r2[None, :] / (2.0 * np.asarray(sigmas)[:, None] ** 2)
    f = -1.5 * np.log(2 * np.pi * np.asarray(sigmas) ** 2)
    return u_kn, N_k, f - f[0]


def test_mbar_recovers_harmonic_oscillator_free_energies_with_honest_errors():
    """The reference's acceptance test (tests/test_sampling.py:100-300: sigma_k = 1 + 0.2 k, first and last state
    unsampled, |error| <= 6 standard errors), on exact samples; the reported standard error of f_last - f_first is
    also compared with its scatter over independent repetitions."""
    sigmas = [1.0 + 0.2 * k for k in range(7)]
    sampled = [1, 2, 3, 4, 5]
    rng = np.random.default_rng(3)
    est, err = [], []
    for rep in range(24):
        u_kn, N_k, f_exact = _harmonic_samples(sigmas, sampled, 300, rng)
        mbar = an.MBAR(u_kn, N_k)
        D, dD = mbar.compute_free_energy_differences()
        exact = f_exact[None, :] - f_exact[:, None]
        nz = dD > 0
        assert np.all(np.abs(D - exact)[nz] / dD[nz] < 6.0)
        assert np.allclose(np.diag(D), 0.0) and np.allclose(D, -D.T)
        est.append(D[0, -1]); err.append(dD[0, -1])
    est, err = np.array(est), np.array(err)
    exact = -1.5 * np.log(sigmas[-1] ** 2 / sigmas[0] ** 2)
    assert abs(est.mean() - exact) < 4.0 * est.std() / np.sqrt(len(est)) + 0.01
    assert 0.6 < err.mean() / est.std(ddof=1) < 1.6                 # asymptotic error vs observed scatter
    # warm start reproduces the same solution
    again = an.MBAR(u_kn, N_k, initial_f_k=mbar.f_k)
    assert np.allclose(again.f_k, mbar.f_k, atol=1e-9)


def test_mbar_two_states_solves_the_bar_equation():
    """K = 2: eq. 11 reduces to Bennett's implicit equation sum_F fermi(M + w_F - df) = sum_R fermi(-M + w_R + df)."""
    rng = np.random.default_rng(4)
    u_kn, N_k, _ = _harmonic_samples([1.0, 1.4], [0, 1], 400, rng)
    df = an.MBAR(u_kn, N_k).f_k[1]
    w_F = (u_kn[1] - u_kn[0])[:400]
    w_R = (u_kn[0] - u_kn[1])[400:]
    fermi = lambda x: 1.0 / (1.0 + np.exp(x))
    M = np.log(400 / 400)
    assert abs(fermi(M + w_F - df).sum() - fermi(-M + w_R + df).sum()) < 1e-6
    with pytest.raises(an.ParameterError):
        an.MBAR(u_kn, [400, 399])


def _pt_sampler(tmp_path, n_iter, **kw):
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=5.0 / unit.picosecond,
                                              n_steps=40, reassign_velocities=True, splitting='V R O R V')
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=n_iter, engine=OracleEngine(), seed=11, **kw)
    rep = MultiStateReporter(str(tmp_path / 'store'), checkpoint_interval=2)
    s.create(ts, [ss], storage=rep, min_temperature=300.0, max_temperature=600.0, n_temperatures=4)
    return s, rep


def test_online_analysis_is_written_and_read_back(tmp_path):
    """tests/test_sampling.py:2213-2296: stored online estimates are current; resuming restores them."""
    s, rep = _pt_sampler(tmp_path, 6, online_analysis_interval=2, online_analysis_minimum_iterations=3)
    s.run()
    f_k, (fe, err) = ParallelTemperingSampler._read_last_free_energy(rep, s.iteration)
    assert len(s._last_mbar_f_k) == 4 and not np.all(s._last_mbar_f_k == 0)
    assert np.all(s._last_mbar_f_k == f_k) and s._last_mbar_f_k[0] == 0.0
    assert fe is not None                       # (the MBAR pass of iteration 6 overwrites the online value, as in the reference)
    assert s._last_err_free_energy != 0 and s._last_err_free_energy == err
    # the stochastic-approximation recursion itself (multistatesampler.py:1636-1657), recomputed from the stored energies
    e, _, _ = rep.read_energies()
    f = np.zeros(4)
    for it in range(1, s.iteration + 1):
        logZ = -f
        for r in range(4):
            lp = -e[it, r] - np.logaddexp.reduce(-e[it, r])
            logZ = logZ + np.exp(lp) / float(it + 1)
        logZ -= logZ[0]
        f = -logZ
    assert np.allclose(f, s._last_mbar_f_k, rtol=1e-12, atol=1e-14)
    resumed = ParallelTemperingSampler.from_storage(rep, engine=OracleEngine())
    assert np.array_equal(resumed._last_mbar_f_k, rep.read_online_analysis_data(resumed.iteration, 'f_k')['f_k'])
    assert resumed.online_analysis_interval == 2 and resumed.online_analysis_minimum_iterations == 3


def test_online_analysis_stops_the_run_at_the_target_error(tmp_path):
    """tests/test_sampling.py:2297-2325: an infinite target error completes the simulation after one iteration."""
    s, _ = _pt_sampler(tmp_path, 5, online_analysis_interval=1, online_analysis_minimum_iterations=0,
                       online_analysis_target_error=np.inf)
    s.run()
    assert s.iteration < 5 and s.is_completed
    with pytest.raises(ValueError):
        ParallelTemperingSampler(online_analysis_interval=0, engine=OracleEngine())
    with pytest.raises(ValueError):
        ParallelTemperingSampler(online_analysis_interval=1, online_analysis_target_error=-1.0, engine=OracleEngine())
    s2, _ = _pt_sampler(tmp_path / 'b', 2, online_analysis_interval=None)
    s2.run()
    assert s2._last_mbar_f_k is None and s2.is_completed


def test_offline_mbar_on_a_parallel_tempering_run_matches_the_analytical_free_energy(tmp_path):
    """Temperature ladder on one harmonic oscillator: f_j - f_i = -3/2 ln(T_j / T_i); the MBAR estimate computed from the
    reporter by ``_offline_analysis`` has to sit within 6 of its own standard errors (tests/test_sampling.py:276-300)."""
    (tmp_path / 'x').mkdir()
    s, rep = _pt_sampler(tmp_path / 'x', 240, online_analysis_interval=120)
    s.run()
    data = rep.read_online_analysis_data(None, 'free_energy', 'f_k_offline')
    fe, err = data['free_energy']
    exact = -1.5 * np.log(600.0 / 300.0)
    assert np.isfinite(err) and 0.0 < err < 0.2
    assert abs(fe - exact) < 6.0 * err, (fe, exact, err)
    assert s._last_err_free_energy == err
    T = np.array([st.temperature for st in s.thermodynamic_states])
    assert np.allclose(data['f_k_offline'], -1.5 * np.log(T / T[0]), atol=6.0 * err + 0.05)
    a = an.MultiStateSamplerAnalyzer(rep)
    n_eq, g, n_eff = a._get_equilibration_data()
    assert n_eq >= 1 and g >= 1.0 and n_eff > 20
    with pytest.raises(Exception, match='Cannot specify statistical_inefficiency without n_equilibration_iterations'):
        an.MultiStateSamplerAnalyzer(rep, statistical_inefficiency=10)
    b = an.MultiStateSamplerAnalyzer(rep, n_equilibration_iterations=10, statistical_inefficiency=3)
    assert b._get_equilibration_data()[:2] == (10, 3)
    import os, yaml
    docs = yaml.safe_load(open(str(tmp_path / 'x' / 'store') + '_real_time_analysis.yaml'))
    assert docs[-1]['iteration'] == 240 and abs(docs[-1]['mbar_analysis']['free_energy_in_kT'] - fe) < 1e-12


def test_offline_mbar_reports_free_energies_relative_to_the_unsampled_end_states(tmp_path):
    """With two unsampled states they take the end points of the MBAR state list (multistateanalyzer.py:1517-1536) and the
    reported free energy is f(last unsampled) - f(first unsampled), here -3/2 ln(700 / 250)."""
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=5.0 / unit.picosecond,
                                              n_steps=40, reassign_velocities=True, splitting='V R O R V')
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=200, engine=OracleEngine(), seed=5,
                                 online_analysis_interval=200)
    rep = MultiStateReporter(str(tmp_path / 'store'), checkpoint_interval=100)
    unsampled = [states.ThermodynamicState(ho.system, T * unit.kelvin) for T in (250.0, 700.0)]
    s.create(ts, [ss], storage=rep, min_temperature=300.0, max_temperature=600.0, n_temperatures=4,
             unsampled_thermodynamic_states=unsampled)
    s.run()
    a = an.MultiStateSamplerAnalyzer(rep)
    u_ln, N_l = a._compute_mbar_decorrelated_energies()
    assert u_ln.shape[0] == 6 and N_l[0] == 0 and N_l[-1] == 0 and N_l[1:-1].sum() == u_ln.shape[1]
    D, dD = a.get_free_energy()
    exact = -1.5 * np.log(700.0 / 250.0)
    assert abs(D[0, -1] - exact) < 6.0 * dD[0, -1] and 0.0 < dD[0, -1] < 0.5
    fe, err = rep.read_online_analysis_data(None, 'free_energy')['free_energy']
    assert fe == D[0, -1] and err == dD[0, -1] and len(rep.read_online_analysis_data(None, 'f_k_offline')['f_k_offline']) == 6


def test_extend_read_status_and_stored_options(tmp_path):
    """multistatesampler.py:307-358 (read_status), :806-822 (extend), :1145-1167 (stored options)."""
    (tmp_path / 'a').mkdir(); (tmp_path / 'b').mkdir()
    s, rep = _pt_sampler(tmp_path / 'a', 2, online_analysis_interval=1)
    assert s.is_periodic is False and repr(s) == '<instance of ParallelTemperingSampler>' and list(s.metadata) == ['title']            # multistatesampler.py:870-877: the default title
    assert s.metadata['title'].startswith('Parallel tempering simulation created using ParallelTempering class')
    s.run()
    assert s.iteration == 2 and s.is_completed
    st = ParallelTemperingSampler.read_status(rep)
    assert st == (2, None, True) and st.iteration == 2 and st.target_error is None
    s.run(3)                                              # run() never passes number_of_iterations ...
    assert s.iteration == 2
    s.extend(2)                                           # ... extend() does, and stores the new limit
    assert s.iteration == 4 and s.number_of_iterations == 4 and s.options['number_of_iterations'] == 4
    assert ParallelTemperingSampler.read_status(str(tmp_path / 'a' / 'store')) == (4, None, True)
    resumed = ParallelTemperingSampler.from_storage(rep, engine=OracleEngine())
    assert resumed.number_of_iterations == 4 and resumed.options == s.options
    # a statistical-error stopping condition shows up in the status
    s2, rep2 = _pt_sampler(tmp_path / 'b', 5, online_analysis_interval=1, online_analysis_target_error=np.inf)
    # nothing estimated yet: the error defaults to inf, which meets an infinite target (the reference's trap, :343-346)
    assert ParallelTemperingSampler.read_status(rep2) == (0, np.inf, True)
    s2.run()
    st2 = ParallelTemperingSampler.read_status(rep2)
    assert st2.iteration == s2.iteration < 5 and st2.target_error == np.inf and st2.is_completed


def test_analyzer_on_a_sams_run_uses_the_expanded_ensemble_timeseries(tmp_path):
    """SAMS storage carries per-iteration logZ / log_weights: the effective-energy timeseries gets the expanded-ensemble
    correction (multistateanalyzer.py:1446-1470) and equilibration starts no earlier than SAMS' second stage (:2068-2076);
    MBAR on the expanded-ensemble samples still has to recover -3/2 ln(T_j / T_i), and SAMS' own logZ should agree."""
    from openmmtools_amd.multistate import SAMSSampler
    ho = testsystems.HarmonicOscillator()
    T = np.linspace(300.0, 500.0, 5)
    sts = [states.ThermodynamicState(ho.system, t * unit.kelvin) for t in T]
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=5.0 / unit.picosecond,
                                              n_steps=40, reassign_velocities=True, splitting='V R O R V')
    s = SAMSSampler(mcmc_moves=move, number_of_iterations=300, engine=OracleEngine(), seed=21,
                    flatness_criteria='minimum-visits', online_analysis_interval=None)
    rep = MultiStateReporter(str(tmp_path / 'store'), checkpoint_interval=50)
    s.create(sts, [ss] * 2, storage=rep)
    s.run()
    a = an.MultiStateSamplerAnalyzer(rep)
    assert a.has_log_weights and a._sams_t0() is not None and a._sams_t0() >= 1
    e, eu, nb, st = a._read_energies()
    plain = np.array([e[np.arange(2), st[:, it], it].sum() for it in range(e.shape[-1])])
    u_n = a.get_effective_energy_timeseries(e, st)
    assert u_n.shape == plain.shape and not np.allclose(u_n, plain)
    n_eq, g, n_eff = a._get_equilibration_data()
    assert n_eq >= a._sams_t0()
    D, dD = a.get_free_energy()
    exact = -1.5 * np.log(T[None, :] / T[:, None])
    assert np.all(np.abs(D - exact)[dD > 0] < 6.0 * dD[dD > 0])
    assert abs((-s._logZ[-1] + s._logZ[0]) - exact[0, -1]) < 0.5          # SAMS' online estimate, loosely
    # a plain parallel-tempering store has no weights
    (tmp_path / 'p').mkdir()
    s2, rep2 = _pt_sampler(tmp_path / 'p', 3, online_analysis_interval=None)
    s2.run()
    assert not an.MultiStateSamplerAnalyzer(rep2).has_log_weights


def test_mixing_statistics_from_the_stored_state_trajectory(tmp_path):
    """multistateanalyzer.py:1243-1303: symmetrised transition counts, eigenvalues in descending order, state-index
    statistical inefficiency."""
    (tmp_path / 'm').mkdir()
    s, rep = _pt_sampler(tmp_path / 'm', 60, online_analysis_interval=None)
    s.run()
    a = an.MultiStateSamplerAnalyzer(rep)
    ms = a.generate_mixing_statistics(number_equilibrated=1)
    T = ms.transition_matrix
    assert T.shape == (4, 4) and np.allclose(T.sum(axis=1), 1.0) and np.all(T >= 0)
    assert np.isclose(ms.eigenvalues[0].real, 1.0) and np.all(np.diff(ms.eigenvalues.real) <= 1e-12)
    assert abs(ms.eigenvalues[1]) < 1.0 and ms.statistical_inefficiency >= 1.0
    states = rep.read_replica_thermodynamic_states()
    n = np.zeros((4, 4))
    for it in range(1, states.shape[0] - 1):
        for r in range(4):
            n[states[it, r], states[it + 1, r]] += 1
    i = 2
    assert np.allclose(T[i], (n[i] + n[:, i]) / (n[i].sum() + n[:, i].sum()))
    # default: the automatically detected equilibration
    assert a.generate_mixing_statistics().transition_matrix.shape == (4, 4)
    # several series of one AR(1) process share one inefficiency
    rng = np.random.default_rng(9)
    series = []
    for _ in range(6):
        x = np.zeros(5000); e = rng.normal(size=5000)
        for t in range(1, 5000):
            x[t] = 0.6 * x[t - 1] + e[t]
        series.append(x)
    g = an.statistical_inefficiency_multiple(series)
    assert abs(g - 4.0) < 0.5


def test_mbar_enthalpy_and_entropy_of_harmonic_oscillators():
    """tests/test_sampling.py:404-437: Delta H and Delta S within 6 standard errors.  For u_k = |x|^2 / (2 sigma_k^2) in
    3-D, <u_k>_k = 3/2 for every k (equipartition), so Delta_u = 0 and Delta_s = -Delta_f; the reported errors are compared
    with the scatter over repetitions, and the free-energy part agrees with compute_free_energy_differences."""
    sigmas = [1.0 + 0.2 * k for k in range(7)]
    sampled = [1, 2, 3, 4, 5]
    rng = np.random.default_rng(12)
    du, ddu, ds, dds = [], [], [], []
    for rep in range(16):
        u_kn, N_k, f_exact = _harmonic_samples(sigmas, sampled, 400, rng)
        mbar = an.MBAR(u_kn, N_k)
        r = mbar.compute_entropy_and_enthalpy()
        D, dD = mbar.compute_free_energy_differences()
        assert np.allclose(r['Delta_f'], D) and np.allclose(r['dDelta_f'], dD, atol=1e-10)
        exact_f = f_exact[None, :] - f_exact[:, None]
        for key, dkey, exact in (('Delta_u', 'dDelta_u', np.zeros((7, 7))), ('Delta_s', 'dDelta_s', -exact_f)):
            nz = r[dkey] > 0
            assert np.all(np.abs(r[key] - exact)[nz] / r[dkey][nz] < 6.0), key
            assert np.allclose(r[key], -r[key].T)
        du.append(r['Delta_u'][1, 5]); ddu.append(r['dDelta_u'][1, 5]); ds.append(r['Delta_s'][1, 5]); dds.append(r['dDelta_s'][1, 5])
    assert 0.5 < np.mean(ddu) / np.std(du, ddof=1) < 2.0
    assert 0.5 < np.mean(dds) / np.std(ds, ddof=1) < 2.0
    # identity Delta_s = Delta_u - Delta_f holds for the estimates themselves
    assert np.allclose(r['Delta_s'], r['Delta_u'] - r['Delta_f'], atol=1e-12)


def test_phase_surface_and_the_combination_of_phases(tmp_path):
    """multistateanalyzer.py:446-1134 (PhaseAnalyzer properties), :2224-2570 (MultiPhaseAnalyzer): ``a - b`` reports the signed
    sum of the phases' free energies between their reference states with the errors added in quadrature."""
    from openmmtools_amd.constants import kB
    runs = []
    for name in ('complex', 'solvent'):
        (tmp_path / name).mkdir()
        s, rep = _pt_sampler(tmp_path / name, 120, online_analysis_interval=None)
        s.run()
        a = an.MultiStateSamplerAnalyzer(rep)
        a.name = name
        runs.append(a)
    a, b = runs
    assert a.n_iterations == 120 and a.n_replicas == a.n_states and abs(a.kT - kB * 300.0) < 1e-9 and a.reporter is not None
    e, eu, nb, st = a.read_energies()
    assert e.shape == (a.n_replicas, a.n_states, 121) and st.shape == (a.n_replicas, 121)
    assert a.effective_length > 5
    text = a.show_mixing_statistics(cutoff=0.01)
    assert 'Perron eigenvalue' in text and 'transition matrix' in text
    fa, da = a.get_free_energy()
    fb, db = b.get_free_energy()
    diff = a - b
    assert isinstance(diff, an.MultiPhaseAnalyzer) and diff.names == ['complex', 'solvent'] and diff.signs == ['+', '-']
    v, err = diff.get_free_energy()
    assert abs(v - (fa[0, -1] - fb[0, -1])) < 1e-12 and abs(err - np.hypot(da[0, -1], db[0, -1])) < 1e-12
    total = a + b
    assert total.signs == ['+', '+'] and abs(total.get_free_energy()[0] - (fa[0, -1] + fb[0, -1])) < 1e-12
    three = diff - a                                            # a name that is taken gets a counter (multistate/utils.py:60-95)
    assert three.names == ['complex', 'solvent', 'complex0'] and three.signs == ['+', '-', '-']
    assert abs(three.get_free_energy()[0] + fb[0, -1]) < 1e-12
    neg = -diff
    assert neg.signs == ['-', '+'] and abs(neg.get_enthalpy()[0] + diff.get_enthalpy()[0]) < 1e-12
    a.reference_states = (0, 1)
    assert abs((a - b).get_free_energy()[0] - (fa[0, 1] - fb[0, -1])) < 1e-12
    a.clear()
    assert a._mbar is None
    with pytest.raises(TypeError):
        diff + 3


def test_helpers_match_vectors_executed_from_the_reference():
    """tests/golden/analysis_reference.json (tests/golden/make_golden_analysis.py): reformat_energies_for_mbar and
    generate_phase_name run from the reference's own source."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'analysis_reference.json')))
    u = np.array(g['u_kln'])
    assert np.array_equal(an.MultiStateSamplerAnalyzer.reformat_energies_for_mbar(u), np.array(g['full']))
    assert np.array_equal(an.MultiStateSamplerAnalyzer.reformat_energies_for_mbar(u, g['ragged_n_k']), np.array(g['ragged']))
    for current, taken, want in g['names']:
        assert an.generate_phase_name(current, taken) == want


def test_online_analysis_works_like_the_references_test(tmp_path):
    """tests/test_sampling.py:2213-2295 with its parameters (10 iterations, analysis every 2, at least 3 iterations, the
    Verlet-integrator move): what the sampler holds after run() is what _read_last_free_energy reads back from the storage."""
    from openmmtools_amd import integrators
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(ho.positions + 0.01, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.SequenceMove([mcmc.IntegratorMove(integrators.VelocityVerletIntegrator(1.0 * unit.femtosecond), n_steps=1),
                              mcmc.LangevinDynamicsMove(timestep=2.0 * unit.femtosecond, n_steps=20, reassign_velocities=True)])
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=10, online_analysis_interval=2, online_analysis_minimum_iterations=3,
                                 engine=OracleEngine(), seed=12)
    rep = MultiStateReporter(str(tmp_path / 'store'), checkpoint_interval=2)
    s.create(ts, [ss], storage=rep, min_temperature=300.0, max_temperature=600.0, n_temperatures=3)
    s.run()
    f_k, (free_energy, err) = ParallelTemperingSampler._read_last_free_energy(s._reporter, s.iteration)
    assert len(s._last_mbar_f_k) == 3 and not np.all(s._last_mbar_f_k == 0)
    assert np.all(s._last_mbar_f_k == f_k) and free_energy is not None
    assert s._last_err_free_energy != 0 and s._last_err_free_energy == err


def _yaml_sampler(tmp_path, n_iterations, online_interval, checkpoint_interval, name='store.nc'):
    from openmmtools_amd import integrators
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(ho.positions + 0.01, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinDynamicsMove(timestep=2.0 * unit.femtosecond, n_steps=5, reassign_velocities=True)
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=n_iterations, online_analysis_interval=online_interval,
                                 engine=OracleEngine(), seed=12)
    rep = MultiStateReporter(str(tmp_path / name), checkpoint_interval=checkpoint_interval)
    s.create(ts, [ss], storage=rep, min_temperature=300.0, max_temperature=600.0, n_temperatures=3)
    return s, rep, str(tmp_path / (os.path.splitext(name)[0] + '_real_time_analysis.yaml'))


def test_real_time_analysis_yaml_has_one_entry_per_analysis(tmp_path):
    """tests/test_sampling.py:2385-2426: 13 iterations, analysis every 3 => int(13 / 3) entries in <name>_real_time_analysis.yaml."""
    import yaml
    s, rep, path = _yaml_sampler(tmp_path, 13, 3, 3)
    s.run()
    assert len(yaml.safe_load(open(path))) == int(13 / 3)


@pytest.mark.parametrize('n_iterations,online_interval,checkpoint_interval,iterations_first_run',
                         [(15, 3, 5, 11), (15, 3, 5, 3), (10, 2, 2, 3), (10, 2, 2, 4), (10, 2, 2, 2)])
def test_real_time_analysis_yaml_after_a_restore(tmp_path, n_iterations, online_interval, checkpoint_interval, iterations_first_run):
    """tests/test_sampling.py:2428-2500 with its five cases: the entries of the first run, then (after from_storage resumes at the
    last checkpoint and repeats the iterations behind it) the total the reference expects."""
    import yaml
    first = iterations_first_run // online_interval
    checkpoints = iterations_first_run // checkpoint_interval
    extra = first - checkpoint_interval * checkpoints // online_interval
    total = n_iterations // online_interval + extra
    s, rep, path = _yaml_sampler(tmp_path, n_iterations, online_interval, checkpoint_interval)
    s.run(n_iterations=iterations_first_run)
    got = yaml.safe_load(open(path)) if os.path.exists(path) else []
    assert len(got or []) == first
    del s
    back = ParallelTemperingSampler.from_storage(rep, engine=OracleEngine())
    back.run()
    assert len(yaml.safe_load(open(path))) == total
