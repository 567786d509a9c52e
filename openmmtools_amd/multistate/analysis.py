"""Free-energy analysis of a stored multistate simulation: what ``MultiStateSampler._offline_analysis`` needs.

The reference runs ``MultiStateSamplerAnalyzer`` (openmmtools/multistate/multistateanalyzer.py:1164) on the
reporter's energies and hands them to **pymbar** (a dependency that is not vendored in the reference tree and not
installed here; ``devtools/conda-envs/test_env.yaml:14`` lists it unpinned, i.e. pymbar 4).  This module restates the
published algorithms the analyzer relies on, in plain numpy:

* ``statistical_inefficiency`` / ``subsample_correlated_data`` — pymbar.timeseries (Chodera et al., JCTC 3, 26 (2007);
  integrated autocorrelation time summed until the normalised fluctuation autocorrelation function first crosses zero,
  ``mintime = 3``, optional ``fast`` increments);
* ``get_equilibration_data_per_sample`` — openmmtools/multistate/utils.py:107-190 (automated equilibration detection of
  Chodera, JCTC 12, 1799 (2016) on at most ``max_subset`` time origins);
* ``MBAR`` — Shirts & Chodera, J. Chem. Phys. 129, 124105 (2008): self-consistent / Newton solution of eq. 11 for the
  dimensionless free energies and the asymptotic covariance of eq. 8 / D6 (SVD form) for their uncertainties;
* ``MultiStateSamplerAnalyzer`` — the analyzer's pipeline for global neighbourhoods: effective-energy timeseries
  (multistateanalyzer.py:1414-1478), equilibration data (:2026-2088), decorrelated u_ln / N_l assembly with the
  unsampled states at the end points (:1479-1545), free energy differences (:1919-2003).

Parity: unpinned against pymbar itself (absent); pinned against the analytical free energies of harmonic oscillators,
the reference's own acceptance test (tests/test_sampling.py:100-300), in tests/test_analysis_cpu.py.
"""
import collections

import numpy as np


# ---- timeseries ------------------------------------------------------------------------------------------------
def statistical_inefficiency(A_n, fast=False, mintime=3):
    """g = 1 + 2 tau of the stationary timeseries ``A_n`` (>= 1).  pymbar.timeseries.statistical_inefficiency."""
    A = np.asarray(A_n, dtype=np.float64)
    N = A.size
    dA = A - A.mean()
    sigma2 = np.mean(dA * dA)
    if sigma2 == 0.0:
        raise ValueError('sample covariance is zero: cannot compute the statistical inefficiency')
    g = 1.0
    t, increment = 1, 1
    while t < N - 1:
        C = np.dot(dA[:N - t], dA[t:]) / (float(N - t) * sigma2)
        if C <= 0.0 and t > mintime:
            break
        g += 2.0 * C * (1.0 - float(t) / float(N)) * float(increment)
        t += increment
        if fast:
            increment += 1
    return max(g, 1.0)


def statistical_inefficiency_multiple(A_kn, fast=False, mintime=3):
    """One statistical inefficiency from several timeseries of the same observable (pymbar.timeseries): the fluctuation
    autocorrelation function is averaged over the series before it is integrated."""
    series = [np.asarray(a, dtype=np.float64) for a in A_kn]
    N_k = np.array([a.size for a in series])
    Nmax, N = int(N_k.max()), int(N_k.sum())
    mu = sum(a.sum() for a in series) / float(N)
    dA = [a - mu for a in series]
    sigma2 = sum((d * d).sum() for d in dA) / float(N)
    if sigma2 == 0.0:
        raise ValueError('sample covariance is zero: cannot compute the statistical inefficiency')
    g = 1.0
    t, increment = 1, 1
    while t < Nmax - 1:
        num, den = 0.0, 0
        for d in dA:
            if t >= d.size:
                continue
            x = d[:d.size - t] * d[t:]
            num += x.sum(); den += x.size
        C = (num / float(den)) / sigma2
        if C <= 0.0 and t > mintime:
            break
        g += 2.0 * C * (1.0 - float(t) / N_k.mean()) * float(increment)
        t += increment
        if fast:
            increment += 1
    return max(g, 1.0)


def subsample_correlated_data(A_t, g=None, fast=False, conservative=False):
    """Indices of an (approximately) uncorrelated subset: every ``g``-th sample, rounded (pymbar.timeseries)."""
    T = len(A_t)
    if g is None:
        g = statistical_inefficiency(A_t, fast=fast)
    if conservative:
        stride = int(np.ceil(g))
        return list(range(0, T, stride))
    indices, n = [], 0
    while int(round(n * g)) < T:
        t = int(round(n * g))
        if n == 0 or t != indices[-1]:
            indices.append(t)
        n += 1
    return indices


def get_equilibration_data_per_sample(timeseries_to_analyze, fast=True, max_subset=100):
    """(i_t, g_i, n_effective_i) on at most ``max_subset`` evenly spaced time origins (multistate/utils.py:107-190)."""
    series = np.array(timeseries_to_analyze)
    time_size = series.size
    set_size = time_size - 1
    if max_subset is None or set_size < max_subset:
        max_subset = set_size
    if max_subset == 0:
        max_subset = 1
    if series.std() == 0.0 or max_subset == 1:
        return (np.arange(max_subset, dtype=int), np.array([1] * max_subset),
                np.arange(time_size, time_size - max_subset, -1))
    g_i = np.ones([max_subset], np.float32)
    n_effective_i = np.ones([max_subset], np.float32)
    counter = np.arange(max_subset)
    i_t = np.floor(counter * time_size / max_subset).astype(int)
    for i, t in enumerate(i_t):
        g_i[i] = statistical_inefficiency(series[t:], fast=fast)       # (:176-181: an error here is raised, not papered over)
        n_effective_i[i] = (time_size - t + 1) / g_i[i]
    return i_t[1:], g_i[1:], n_effective_i[1:]                          # :190: the origin t = 0 is not a candidate


def get_decorrelation_time(timeseries_to_analyze):
    """multistate/utils.py:98-104."""
    return statistical_inefficiency(timeseries_to_analyze)


def get_equilibration_data(timeseries_to_analyze, fast=True, max_subset=1000):
    """multistate/utils.py:195-235 (kept by the reference for old callers): (n_equilibration, g_t, n_effective_max)."""
    i_t, g_i, n_effective_i = get_equilibration_data_per_sample(timeseries_to_analyze, fast=fast, max_subset=max_subset)
    i_max = n_effective_i.argmax()
    return i_t[i_max], g_i[i_max], n_effective_i.max()


def remove_unequilibrated_data(data, number_equilibrated, axis):
    """multistate/utils.py:238-266: drop the first ``number_equilibrated`` entries along ``axis``."""
    cast = np.asarray(data)
    slc = [slice(None)] * cast.ndim
    slc[axis] = slice(number_equilibrated, None)
    return cast[tuple(slc)]


def subsample_data_along_axis(data, subsample_rate, axis):
    """multistate/utils.py:269-300: every ``subsample_rate``-th entry (rounded as subsample_correlated_data does) along ``axis``."""
    cast = np.asarray(data)
    indices = subsample_correlated_data(np.zeros(cast.shape[axis]), g=subsample_rate)
    return np.take(cast, indices, axis=axis)


def generate_phase_name(current_name, name_list):
    """multistate/utils.py:60-95: a name not yet in ``name_list`` ('phase0', 'phase1', ... or the given name + counter)."""
    counter = 0
    if current_name is None:
        name = 'phase%d' % counter
        while name in name_list:
            counter += 1
            name = 'phase%d' % counter
        return name
    if current_name in name_list:
        name = current_name + str(counter)
        while name in name_list:
            counter += 1
            name = current_name + str(counter)
        return name
    return current_name


# ---- MBAR --------------------------------------------------------------------------------------------------------
def _logsumexp(a, axis=None, b=None):
    a = np.asarray(a, dtype=np.float64)
    amax = np.max(a, axis=axis, keepdims=True)
    amax = np.where(np.isfinite(amax), amax, 0.0)
    e = np.exp(a - amax)
    if b is not None:
        e = e * b
    s = np.sum(e, axis=axis, keepdims=True)
    out = np.log(s) + amax
    return np.squeeze(out, axis=axis) if axis is not None else float(out.reshape(()))


class ParameterError(Exception):
    """Raised when the estimator cannot be computed from the data it was given (pymbar.utils.ParameterError)."""


class MBAR:
    """Multistate Bennett acceptance ratio estimator.

    ``u_kn[k, n]`` is the reduced potential of sample ``n`` evaluated in state ``k``; ``N_k[k]`` the number of samples
    drawn from state ``k`` (0 for unsampled states).  Solves  f_i = -ln sum_n exp(-u_in) / sum_k N_k exp(f_k - u_kn)
    (eq. 11) with f_0 = 0.
    """

    def __init__(self, u_kn, N_k, initial_f_k=None, relative_tolerance=1.0e-12, maximum_iterations=10000):
        self.u_kn = np.array(u_kn, dtype=np.float64)
        self.N_k = np.array(N_k, dtype=np.int64)
        K, N = self.u_kn.shape
        if self.N_k.shape != (K,) or int(self.N_k.sum()) != N:
            raise ParameterError('N_k must have one entry per state and sum to the number of samples')
        if not np.all(np.isfinite(self.u_kn[self.N_k > 0])):
            raise ParameterError('non-finite reduced potentials in a sampled state')
        self.K, self.N = K, N
        f = np.zeros(K) if initial_f_k is None else np.array(initial_f_k, dtype=np.float64) - float(np.asarray(initial_f_k)[0])
        self.f_k = self._solve(f, relative_tolerance, maximum_iterations)
        self._log_W = None

    # log of the mixture denominator per sample: ln sum_k N_k exp(f_k - u_kn)
    def _log_denominator(self, f_k):
        s = self.N_k > 0
        return _logsumexp(f_k[s, None] - self.u_kn[s, :], axis=0, b=self.N_k[s, None].astype(np.float64))

    def _solve(self, f_k, rtol, maxiter):
        s = np.flatnonzero(self.N_k > 0)
        if s.size == 0:
            raise ParameterError('no sampled state')
        u = self.u_kn[s]
        Nk = self.N_k[s].astype(np.float64)
        f = f_k[s] - f_k[s][0]

        def objective_parts(f):
            log_den = _logsumexp(f[:, None] - u, axis=0, b=Nk[:, None])
            log_W = f[:, None] - u - log_den[None, :]                     # [Ks, N]
            return log_den, log_W

        # a few self-consistent sweeps bring any starting point into Newton's basin
        for _ in range(5):
            log_den, _ = objective_parts(f)
            f_new = -_logsumexp(-u - log_den[None, :], axis=1)
            f = f_new - f_new[0]
        for _ in range(maxiter):
            log_den, log_W = objective_parts(f)
            W = np.exp(log_W)
            g = Nk * (W.sum(axis=1) - 1.0)                                # gradient of the convex objective
            H = np.diag(Nk * W.sum(axis=1)) - (Nk[:, None] * Nk[None, :]) * (W @ W.T)
            if s.size > 1:
                try:
                    step = np.zeros_like(f)
                    step[1:] = np.linalg.solve(H[1:, 1:], g[1:])
                except np.linalg.LinAlgError:
                    step = np.zeros_like(f)
                    step[1:] = np.linalg.lstsq(H[1:, 1:], g[1:], rcond=None)[0]
            else:
                step = np.zeros_like(f)
            phi0 = log_den.sum() - np.dot(Nk, f)
            scale = 1.0
            while True:                                                   # backtrack: the objective must not increase
                f_try = f - scale * step
                phi = objective_parts(f_try)[0].sum() - np.dot(Nk, f_try)
                if phi <= phi0 + 1e-12 * abs(phi0) or scale < 1e-6:
                    break
                scale *= 0.5
            delta = np.max(np.abs(f_try - f))
            f = f_try
            if delta <= rtol * max(1.0, np.max(np.abs(f))):
                break
        else:
            raise ParameterError('MBAR did not converge')
        if not np.all(np.isfinite(f)):
            raise ParameterError('MBAR produced non-finite free energies')
        out = np.zeros(self.K)
        out[s] = f
        # unsampled states: one application of eq. 11 with the converged denominator
        log_den = _logsumexp(f[:, None] - u, axis=0, b=Nk[:, None])
        uns = np.flatnonzero(self.N_k == 0)
        if uns.size:
            out[uns] = -_logsumexp(-self.u_kn[uns] - log_den[None, :], axis=1)
        return out - out[0]

    @property
    def log_W_nk(self):
        if self._log_W is None:
            log_den = self._log_denominator(self.f_k)
            self._log_W = (self.f_k[:, None] - self.u_kn - log_den[None, :]).T          # [N, K]
        return self._log_W

    @staticmethod
    def _theta_of(W, N_k):
        """Asymptotic covariance (eq. 8), SVD form:  Theta = V S (I - S V^T N V S)^+ S V^T  with W = U S V^T — only
        the Gram matrix W^T W is needed.  ``W`` may carry extra columns with N_k = 0 (unsampled or observable-weighted
        states)."""
        G = W.T @ W
        evals, V = np.linalg.eigh(G)
        evals = np.clip(evals, 0.0, None)
        S = np.sqrt(evals)
        M = np.eye(W.shape[1]) - (S[:, None] * (V.T @ (np.asarray(N_k, dtype=np.float64)[:, None] * V))) * S[None, :]
        return (V * S[None, :]) @ np.linalg.pinv(M, rcond=1e-10) @ (V * S[None, :]).T

    def _theta(self):
        return self._theta_of(np.exp(self.log_W_nk), self.N_k)

    def compute_entropy_and_enthalpy(self):
        """Reduced enthalpy <u_i>_i and entropy s_i = <u_i>_i - f_i differences with their uncertainties
        (pymbar's compute_entropy_and_enthalpy; section IV of the MBAR paper: each <u_i>_i is the ratio of the
        normalisation constants of an extra, u-weighted state and of state i, so its error follows from the covariance
        of the augmented set of free energies).  Returns a dict with Delta_f, dDelta_f, Delta_u, dDelta_u, Delta_s,
        dDelta_s; Delta_x[i, j] = x_j - x_i."""
        K = self.K
        log_W = self.log_W_nk                                         # [N, K]
        u = self.u_kn.T                                               # [N, K]
        shift = u.min() - 1.0                                         # observable made positive for the logarithm
        A = u - shift
        log_WA_raw = log_W + np.log(A)
        log_cA = _logsumexp(log_WA_raw, axis=0)                       # ln <A>_i (shifted)
        W_aug = np.concatenate([np.exp(log_W), np.exp(log_WA_raw - log_cA[None, :])], axis=1)
        Theta = self._theta_of(W_aug, np.concatenate([self.N_k, np.zeros(K, dtype=np.int64)]))
        A_i = np.exp(log_cA)
        u_i = A_i + shift
        # x = (f_0..f_{K-1}, u_0..u_{K-1}) is linear in the augmented free energies: du_i = <A>_i (df_i - df_{K+i})
        J = np.zeros((2 * K, 2 * K))
        J[:K, :K] = np.eye(K)
        J[K:, :K] = np.diag(A_i)
        J[K:, K:] = -np.diag(A_i)
        C = J @ Theta @ J.T

        def diff_and_err(value, L):                                   # value_j - value_i and its error; L maps x -> value
            cov = L @ C @ L.T
            d2 = np.diag(cov)[:, None] + np.diag(cov)[None, :] - 2.0 * cov
            d2 = np.where(np.abs(d2) < 1e-14, 0.0, d2)
            with np.errstate(invalid='ignore'):
                err = np.sqrt(d2)
            np.fill_diagonal(err, 0.0)
            return value[None, :] - value[:, None], err
        Lf = np.concatenate([np.eye(K), np.zeros((K, K))], axis=1)
        Lu = np.concatenate([np.zeros((K, K)), np.eye(K)], axis=1)
        Df, dDf = diff_and_err(self.f_k, Lf)
        Du, dDu = diff_and_err(u_i, Lu)
        Ds, dDs = diff_and_err(u_i - self.f_k, Lu - Lf)
        return dict(Delta_f=Df, dDelta_f=dDf, Delta_u=Du, dDelta_u=dDu, Delta_s=Ds, dDelta_s=dDs)

    def compute_free_energy_differences(self):
        """(Delta_f_ij, dDelta_f_ij) with Delta_f_ij[i, j] = f_j - f_i (pymbar's convention)."""
        f = self.f_k
        Delta = f[None, :] - f[:, None]
        Theta = self._theta()
        d2 = np.diag(Theta)[:, None] + np.diag(Theta)[None, :] - 2.0 * Theta
        d2 = np.where(np.abs(d2) < 1e-14, 0.0, d2)
        with np.errstate(invalid='ignore'):
            dDelta = np.sqrt(d2)                     # NaN where the estimate of the variance is negative (under-sampling)
        np.fill_diagonal(dDelta, 0.0)
        return Delta, dDelta


# ---- the analyzer pipeline (global neighbourhoods) --------------------------------------------------------------
class MultiStateSamplerAnalyzer:
    """Free energies from a ``MultiStateReporter`` (multistateanalyzer.py:1164; global neighbourhoods only)."""

    def __init__(self, reporter, n_equilibration_iterations=None, statistical_inefficiency=None, max_subset=100,
                 max_n_iterations=None, use_full_trajectory=False, analysis_kwargs=None):
        if statistical_inefficiency is not None and n_equilibration_iterations is None:
            raise Exception('Cannot specify statistical_inefficiency without n_equilibration_iterations, because '
                            'otherwise n_equilibration_iterations cannot be computed for the given '
                            'statistical_inefficiency.')                                     # :1209-1213
        self._reporter = reporter
        self._n_equilibration_iterations = n_equilibration_iterations
        self._statistical_inefficiency = statistical_inefficiency
        self._max_subset = max_subset
        self._max_n_iterations = max_n_iterations
        self.use_full_trajectory = use_full_trajectory
        self._kwargs = dict(analysis_kwargs or {})
        self._equilibration_data = None
        self._mbar = None
        self.name = None                       # :611-617: the phase name used when analyzers are combined
        self.reference_states = (0, -1)         # :633-643: the two states a phase's free energy is reported between
        self._sign = '+'

    # ---- the PhaseAnalyzer surface (multistateanalyzer.py:446-1134) ----------------------------------------------
    observables = ('free_energy', 'entropy', 'enthalpy')     # default registry (:304-340): two-state observables, errors in quadrature

    @property
    def reporter(self):
        return self._reporter

    @property
    def n_iterations(self):
        """:646-652: the last completely written iteration (capped by ``max_n_iterations``)."""
        last = self._reporter.read_last_iteration(last_checkpoint=False)
        return last if self._max_n_iterations is None else min(last, self._max_n_iterations)

    @property
    def n_replicas(self):
        return int(self._read_energies()[0].shape[0])

    @property
    def n_states(self):
        return int(self._read_energies()[0].shape[1])

    @property
    def kT(self):
        """:684-693: kT of the first stored thermodynamic state, kJ/mol."""
        from .. import constants
        st = self._reporter.read_thermodynamic_states()[0][0]
        return constants.kB * float(st.temperature)

    def read_energies(self):
        """:831-896: (sampled [replica, state, iteration], unsampled, neighborhoods, replica state indices [replica, iteration])."""
        return self._read_energies()

    @staticmethod
    def reformat_energies_for_mbar(u_kln, n_k=None):
        """:993-1040: energies [sampled state k, evaluated state l, sample n] -> the [l, all samples] layout MBAR takes, the first
        ``n_k[k]`` samples of every k laid side by side in the order of k."""
        u_kln = np.asarray(u_kln)
        k, l, n = u_kln.shape
        n_k = np.full(k, n, dtype=np.int64) if n_k is None else np.asarray(n_k, dtype=np.int64)
        return np.concatenate([u_kln[i, :, :n_k[i]] for i in range(k)], axis=1).astype(np.float64) if k else np.zeros([l, 0])

    def clear(self):
        """:596-608 / :1230-1241: forget everything derived from the storage."""
        self._equilibration_data = None
        self._mbar = None

    @property
    def effective_length(self):
        """:2207-2221: number of uncorrelated production samples."""
        e, _, _, states = self._read_energies()
        return self._get_equilibration_data(e, states)[2]

    def show_mixing_statistics(self, cutoff=0.05, number_equilibrated=None):
        """:1305-1351: log the state-to-state transition matrix (entries below ``cutoff`` blank) and what its second
        eigenvalue says about the equilibration of the state walk; returns the text."""
        import logging
        stats = self.generate_mixing_statistics(number_equilibrated=number_equilibrated)
        t_ij, mu = stats.transition_matrix, stats.eigenvalues
        n = t_ij.shape[0]
        lines = ['Cumulative symmetrized state mixing transition matrix:', '%6s' % '' + ''.join('%6d' % j for j in range(n))]
        for i in range(n):
            lines.append('%-6d' % i + ''.join(('%6.3f' % p) if p >= cutoff else '%6s' % '' for p in t_ij[i]))
        if mu[1] >= 1:
            lines.append('Perron eigenvalue is unity; Markov chain is decomposable.')
        elif mu[1] <= 0:
            lines.append('Perron eigenvalue is %9.5f; state equilibration timescale is less than one iteration.' % mu[1].real)
        else:
            lines.append('Perron eigenvalue is %9.5f; state equilibration timescale is ~ %.1f iterations' % (mu[1].real, 1.0 / (1.0 - mu[1].real)))
        text = '\n'.join(lines)
        logging.getLogger(__name__).info(text)
        return text

    def _combine(self, other, operator):
        mine = dict(phases=[self], names=[generate_phase_name(self.name, [])], signs=[self._sign])
        self._sign = '+'
        return MultiPhaseAnalyzer(mine)._combine_phases(other, operator)

    def __add__(self, other):
        return self._combine(other, '+')

    def __sub__(self, other):
        return self._combine(other, '-')

    def __neg__(self):
        self._sign = '-' if self._sign == '+' else '+'           # :1128-1134
        return self

    def _read_energies(self):
        """[replica, state, iteration] arrays like the reference's ``_read_energies`` (:1353-1412)."""
        e, nb, eu = self._reporter.read_energies()
        states = self._reporter.read_replica_thermodynamic_states()
        # records beyond the last completely written iteration are stale (a resume from an earlier checkpoint leaves the
        # abandoned branch's records behind): the reference caps at the last good iteration (multistateanalyzer.py:1353-1412)
        last = self._reporter.read_last_iteration(last_checkpoint=False)
        n = e.shape[0] if last is None else min(e.shape[0], int(last) + 1)
        if self._max_n_iterations is not None:
            n = min(n, self._max_n_iterations + 1)
        return (np.moveaxis(e[:n], 0, -1), np.moveaxis(eu[:n], 0, -1), np.moveaxis(nb[:n], 0, -1),
                np.moveaxis(states[:n], 0, -1))

    @property
    def has_log_weights(self):
        """True when the storage holds per-iteration logZ / log_weights (SAMS), :898-909."""
        try:
            self._reporter.read_online_analysis_data(0, 'logZ', 'log_weights')
            return True
        except (ValueError, IndexError, KeyError):
            return False

    def get_effective_energy_timeseries(self, energies, replica_state_indices):
        """u_n[iteration] = sum over replicas of the reduced potential in the state the replica occupies (:1462-1472);
        with SAMS weights, the expanded-ensemble correction -sum log_w[state] + logsumexp(-f + log_w) per iteration
        (:1446-1470, f = the last online logZ estimate)."""
        n_replicas, _, n_iterations = energies.shape
        u_n = np.zeros([n_iterations], np.float64)
        rep = np.arange(n_replicas)
        log_weights = f_l = None
        if self.has_log_weights:
            lw = self._reporter._files['log_weights'].read()          # [iteration, state]
            if lw.shape[0] >= n_iterations:
                log_weights = lw[:n_iterations].T
                f_l = -self._reporter.read_online_analysis_data(None, 'logZ')['logZ']
        for it in range(n_iterations):
            states = replica_state_indices[:, it]
            u_n[it] = np.sum(energies[rep, states, it])
            if log_weights is not None:
                u_n[it] += -np.sum(log_weights[states, it]) + _logsumexp(-f_l + log_weights[:, it])
        return u_n

    def _get_equilibration_data(self, energies=None, replica_state_indices=None):
        """(n_equilibration_iterations, statistical_inefficiency, n_effective_max), :2026-2088."""
        if energies is None:
            energies, _, _, replica_state_indices = self._read_energies()
        if self._n_equilibration_iterations is not None and self._statistical_inefficiency is not None:
            n_eq, g_t = self._n_equilibration_iterations, self._statistical_inefficiency
            n_eff = (energies.shape[-1] - 1 - n_eq + 1) / g_t
        else:
            u_n = self.get_effective_energy_timeseries(energies, replica_state_indices)
            t0 = self._n_equilibration_iterations if self._n_equilibration_iterations is not None else 1   # drop iteration 0
            t0_sams = self._sams_t0()                                     # :2068-2076: only the asymptotically optimal SAMS stage
            if t0_sams is not None:
                t0 = max(t0, t0_sams)
            i_t, g_i, n_effective_i = get_equilibration_data_per_sample(u_n[t0:], max_subset=self._max_subset)
            n_eff = n_effective_i.max()
            i_max = n_effective_i.argmax()
            n_eq = int(i_t[i_max] + t0)
            g_t = self._statistical_inefficiency if self._statistical_inefficiency is not None else float(g_i[i_max])
        self._equilibration_data = (n_eq, g_t, n_eff)
        return self._equilibration_data

    def _sams_t0(self):
        """Start of SAMS' second stage when the storage records one (written with the online data at checkpoints)."""
        try:
            last = self._reporter.read_last_iteration(last_checkpoint=True)
            data = self._reporter.read_online_data_if_present(last) if last is not None else None
            st = (data or {}).get('sams_state')
            return int(st['t0']) if st and st.get('stage', 0) == 1 and st.get('t0', 0) > 0 else None
        except Exception:
            return None

    @property
    def n_equilibration_iterations(self):
        return (self._equilibration_data or self._get_equilibration_data())[0]

    @property
    def statistical_inefficiency(self):
        return (self._equilibration_data or self._get_equilibration_data())[1]

    def _compute_mbar_decorrelated_energies(self):
        """(u_ln, N_l) with the unsampled states at the two end points (:1479-1545)."""
        e, eu, nb, states = self._read_energies()
        n_eq, g_t, _ = self._get_equilibration_data(e, states)
        if not self.use_full_trajectory:
            e, eu, states = e[..., n_eq:], eu[..., n_eq:], states[..., n_eq:]
            idx = subsample_correlated_data(np.zeros(e.shape[-1]), g=g_t)
            e, eu, states = e[..., idx], eu[..., idx], states[..., idx]
        n_replicas, n_sampled, n_it = e.shape
        n_uns = eu.shape[1]
        n_total = n_sampled + n_uns
        first = int(n_uns / 2.0)
        last = n_total - first
        u_ln = np.zeros([n_total, n_it * n_replicas])
        N_l = np.zeros([n_total], dtype=int)
        u_ln[first:last, :] = np.concatenate([e[k] for k in range(n_replicas)], axis=1)      # kln' -> ln (:1027-1034)
        uniq, counts = np.unique(states, return_counts=True)
        N_l[first:last][uniq] = counts
        if n_uns > 0:
            u_ln[[0, -1], :] = np.concatenate([eu[k] for k in range(n_replicas)], axis=1)
        return u_ln, N_l

    @property
    def mbar(self):
        if self._mbar is None:
            u_ln, N_l = self._compute_mbar_decorrelated_energies()
            self._mbar = MBAR(u_ln, N_l, initial_f_k=self._kwargs.get('initial_f_k'))
        return self._mbar

    MixingStatistics = collections.namedtuple('MixingStatistics', ['transition_matrix', 'eigenvalues', 'statistical_inefficiency'])

    def generate_mixing_statistics(self, number_equilibrated=None):
        """(transition_matrix, eigenvalues sorted from greatest to least, statistical inefficiency of the replicas' state
        indices) from the stored state trajectory (:1243-1303; symmetrised empirical transition counts)."""
        if number_equilibrated is None:
            number_equilibrated = self.n_equilibration_iterations
        states = self._reporter.read_replica_thermodynamic_states()
        if self._max_n_iterations is not None:
            states = states[:self._max_n_iterations + 1]
        n_iter, n_replicas = states.shape
        n_states = int(self._reporter._meta['n_states'])
        n_ij = np.zeros([n_states, n_states], np.int64)
        for it in range(number_equilibrated, n_iter - 1):
            np.add.at(n_ij, (states[it], states[it + 1]), 1)
        t_ij = np.zeros([n_states, n_states], np.float64)
        for i in range(n_states):
            den = float(n_ij[i, :].sum() + n_ij[:, i].sum())
            if den > 0:
                t_ij[i, :] = (n_ij[i, :] + n_ij[:, i]) / den
            else:
                t_ij[i, i] = 1.0
        mu = -np.sort(-np.linalg.eigvals(t_ij))
        g = statistical_inefficiency_multiple(np.transpose(states[number_equilibrated:]))
        return self.MixingStatistics(transition_matrix=t_ij, eigenvalues=mu, statistical_inefficiency=g)

    def get_enthalpy(self):
        """(Delta_u_ij, dDelta_u_ij) of the reduced enthalpy <u_i>_i, :1988-2005."""
        r = self.mbar.compute_entropy_and_enthalpy()
        return r['Delta_u'], r['dDelta_u']

    def get_entropy(self):
        """(Delta_s_ij, dDelta_s_ij) of the reduced entropy s_i = <u_i>_i - f_i, :2007-2024."""
        r = self.mbar.compute_entropy_and_enthalpy()
        return r['Delta_s'], r['dDelta_s']

    def get_free_energy(self):
        """(Delta_f_ij, dDelta_f_ij) in kT between all (unsampled + sampled) states (:1958-2003)."""
        return self.mbar.compute_free_energy_differences()



class ReplicaExchangeAnalyzer(MultiStateSamplerAnalyzer):
    """replicaexchange.py:427-439."""


class ParallelTemperingAnalyzer(ReplicaExchangeAnalyzer):
    """paralleltempering.py:240-252."""


class SAMSAnalyzer(MultiStateSamplerAnalyzer):
    """sams.py:694-704."""


class MultiPhaseAnalyzer:
    """multistateanalyzer.py:2224-2570: several phases combined with signs (``complex - solvent``); an observable of the
    combination is the signed sum of each phase's value between its ``reference_states``, errors added in quadrature (the
    default registry's free_energy / entropy / enthalpy, :304-340)."""

    def __init__(self, phases):
        self._phases, self._names, self._signs = list(phases['phases']), list(phases['names']), list(phases['signs'])
        shared = [o for o in MultiStateSamplerAnalyzer.observables if all(o in getattr(p, 'observables', ()) for p in self._phases)]
        if not shared:
            raise RuntimeError('There are no shared computable observable between the phases, combining them will do nothing.')
        self._observables = tuple(shared)
        for name in shared:
            setattr(self, 'get_' + name, (lambda n: (lambda: self._compute_observable(n)))(name))

    observables = property(lambda self: self._observables)
    phases = property(lambda self: self._phases)
    names = property(lambda self: self._names)
    signs = property(lambda self: self._signs)

    def clear(self):
        for phase in self._phases:
            phase.clear()

    def _combine_phases(self, other, operator='+'):
        phases, names, signs = list(self._phases), list(self._names), list(self._signs)
        flip = lambda sign: '-' if ((operator == '-') != (sign == '-')) else '+'
        if isinstance(other, MultiPhaseAnalyzer):
            for phase, name, sign in zip(other.phases, other.names, other.signs):
                names.append(generate_phase_name(name, [n for n in other.names if n != name] + names))
                signs.append(flip(sign))
                phases.append(phase)
        elif isinstance(other, MultiStateSamplerAnalyzer):
            names.append(generate_phase_name(other.name, names))
            signs.append(flip(other._sign))
            other._sign = '+'
            phases.append(other)
        else:
            raise TypeError("cannot %s 'MultiPhaseAnalyzer' and '%s' objects" % ('add' if operator == '+' else 'subtract', type(other)))
        return MultiPhaseAnalyzer(dict(phases=phases, names=names, signs=signs))

    def __add__(self, other):
        return self._combine_phases(other, '+')

    def __sub__(self, other):
        return self._combine_phases(other, '-')

    def __neg__(self):
        import copy
        out = copy.copy(self)
        out._signs = ['-' if s == '+' else '+' for s in self._signs]
        return out

    def __str__(self):
        return 'MultiPhaseAnalyzer <' + ' '.join('%s%s' % (s, n) for s, n in zip(self._signs, self._names)) + '>'

    def _compute_observable(self, name):
        value, error = 0.0, 0.0
        for phase, sign in zip(self._phases, self._signs):
            v, e = getattr(phase, 'get_' + name)()
            if not isinstance(phase, MultiPhaseAnalyzer):
                i, j = phase.reference_states
                v, e = v[i, j], e[i, j]
            value = value + v if sign == '+' else value - v
            error = (error ** 2 + e ** 2) ** 0.5
        return value, error
