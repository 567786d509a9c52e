"""SAMSSampler: self-adjusted mixture sampling with the global-jump state update on the device.

Mirrors openmmtools/multistate/sams.py (class :43): options (:168-236), ``_pre_write_create``
(:301-372), ``_mix_replicas`` (:395-437), ``_global_jump`` (:477-501, device kernel
sams_global_jump in csrc/mix.hip), ``_update_stage`` (:564-604), ``_update_logZ_estimates``
(:606-681) and ``_update_log_weights`` (:683-691).  The logZ recursion acts on <= K doubles in
replica order and stays on the host in f64.  'local-jump' / 'restricted-range-jump' are
disabled by the reference's own validator (:241-246) and are not offered.
"""
import numpy as np
from .replicaexchange import ReplicaExchangeSampler
from .multistatesampler import MultiStateSampler


class SAMSSampler(ReplicaExchangeSampler):
    _TITLE_TEMPLATE = 'Self-adjusted mixture sampling (SAMS) simulation using SAMSSampler class of openmmtools_amd.multistate on {}'

    def __init__(self, number_of_iterations=1, log_target_probabilities=None, state_update_scheme='global-jump',
                 locality=5, update_stages='two-stage', flatness_criteria='logZ-flatness', flatness_threshold=0.2,
                 weight_update_method='rao-blackwellized', adapt_target_probabilities=False, gamma0=1.0,
                 logZ_guess=None, **kwargs):
        kwargs.pop('replica_mixing_scheme', None)
        super().__init__(number_of_iterations=number_of_iterations, replica_mixing_scheme=None, **kwargs)
        # sams.py:237-278: one validator per option, all with the same sentence (tests/test_sampler_cpu.py holds the texts to the
        # reference's validators executed from its source)
        for value, supported in ((state_update_scheme, ['global-jump']), (update_stages, ['one-stage', 'two-stage']),
                                 (flatness_criteria, ['minimum-visits', 'logZ-flatness', 'histogram-flatness']),
                                 (weight_update_method, ['optimal', 'rao-blackwellized']), (adapt_target_probabilities, [False])):
            if value not in supported:
                raise ValueError("Unknown update scheme '{}'. Supported values are {}.".format(value, supported))
        self.log_target_probabilities = log_target_probabilities
        self.state_update_scheme = state_update_scheme
        self.locality = None                      # global-jump forces global neighbourhoods (:338-339)
        self.update_stages = update_stages
        self.flatness_criteria = flatness_criteria
        self.flatness_threshold = flatness_threshold
        self.weight_update_method = weight_update_method
        self.adapt_target_probabilities = adapt_target_probabilities          # sams.py:287 (only False passes its validator)
        self.gamma0 = gamma0
        self.logZ_guess = logZ_guess
        self._cached_state_histogram = None
        self._gamma = None

    def _ctor_kwargs(self):
        return dict(log_target_probabilities=self.log_target_probabilities, state_update_scheme=self.state_update_scheme,
                    update_stages=self.update_stages, flatness_criteria=self.flatness_criteria,
                    flatness_threshold=self.flatness_threshold, weight_update_method=self.weight_update_method,
                    gamma0=self.gamma0, logZ_guess=self.logZ_guess)

    def _online_data(self):
        """sams.py:618, :681: logZ and log_weights go to storage every iteration; stage bookkeeping with them."""
        # sams.py:618: the stored log_weights of iteration n are the ones the mix of iteration n USED (written before the
        # update); logZ is the post-update estimate (:381-383) and is what a resume rebuilds the weights from
        used = getattr(self, '_log_weights_used', None)
        return dict(logZ=self._logZ, log_weights=self.log_weights if used is None else used,
                    sams_state=dict(stage=self._stage, t0=self._t0, histogram=self._cached_state_histogram.copy(),
                                    iteration=self._iteration))

    def _restore_online(self, data):
        if not data:
            return
        self._logZ = np.array(data['logZ'], np.float64)
        self._update_log_weights()
        st = data.get('sams_state')
        if st and st.get('iteration') == self._iteration:
            self._stage, self._t0 = st['stage'], st['t0']
            if st.get('histogram') is not None:
                self._cached_state_histogram = np.array(st['histogram'])

    def _initialize_stage(self):
        """sams.py:291-296."""
        self._t0 = 0
        self._stage = 1 if self.update_stages == 'one-stage' else 0

    def _pre_write_create(self, thermodynamic_states, sampler_states, storage, **kwargs):
        """sams.py:301-372: replicas = sampler states (NOT tiled to the number of states)."""
        MultiStateSampler._pre_write_create(self, thermodynamic_states, sampler_states, storage, **kwargs)
        self._initialize_stage()
        if self.log_target_probabilities is None:
            self.log_target_probabilities = np.zeros([self.n_states], np.float64) - np.log(self.n_states)
        self.log_target_probabilities = np.array(self.log_target_probabilities, np.float64)
        self._logZ = np.zeros([self.n_states], np.float64)
        if self.logZ_guess is not None:
            if len(self.logZ_guess) != self.n_states:
                raise Exception('Initial logZ_guess (dim {}) must have same number of states as n_states ({})'.format(
                    len(self.logZ_guess), self.n_states))
            self._logZ = np.array(self.logZ_guess, np.float64)
        self._update_log_weights()
        self._cached_state_histogram = np.zeros(self.n_states, dtype=int)

    @property
    def _state_histogram(self):
        return self._cached_state_histogram

    def _report_iteration(self):
        """sams.py:381-393: histogram of visited states, counted at report time."""
        states, counts = np.unique(self._replica_thermodynamic_states, return_counts=True)
        self._cached_state_histogram[states] += counts
        super()._report_iteration()

    def _mix_replicas(self, rng_iteration=None):
        """sams.py:395-437."""
        it = self._iteration if rng_iteration is None else rng_iteration
        K = self.n_states
        self._log_weights_used = np.array(self.log_weights, np.float64)
        labels, nacc, nprop = self._device_mix('sams-global-jump', it, log_weights=self.log_weights)
        self._n_accepted_matrix[:, :] = nacc[:K, :K]
        self._n_proposed_matrix[:, :] = nprop[:K, :K]
        replicas_log_P_k = self._last_log_P
        # the reference updates logZ with the NEW labels (:626) and only outside equilibration (:428)
        self._replica_thermodynamic_states = labels
        if self._iteration > 0 and rng_iteration is None:
            self._update_logZ_estimates(replicas_log_P_k)
            self._update_log_weights()
        return labels

    def _update_stage(self):
        """sams.py:564-604."""
        minimum_visits = 1
        N_k = self._state_histogram
        if self.update_stages == 'two-stage' and self._stage == 0:
            advance = False
            if N_k.sum() == 0:
                return
            if self.flatness_criteria == 'minimum-visits':
                advance = bool(np.all(N_k >= minimum_visits))
            elif self.flatness_criteria == 'histogram-flatness':
                empirical = N_k / N_k.sum()
                pi_k = np.exp(self.log_target_probabilities)
                advance = bool(np.all(np.abs(pi_k - empirical) / pi_k < self.flatness_threshold))
            elif self.flatness_criteria == 'logZ-flatness':
                advance = bool(np.all(np.abs(self._logZ / self.gamma0) > self.flatness_threshold))
            if advance or (self._t0 > 0 and self._iteration > self._t0):
                self._stage = 1
                self._t0 = self._iteration - 1

    def _update_logZ_estimates(self, replicas_log_P_k):
        """sams.py:606-681."""
        log_pi_k = self.log_target_probabilities
        pi_k = np.exp(log_pi_k)
        self._update_stage()
        gamma = None
        for replica_index, state_index in enumerate(self._replica_thermodynamic_states):
            beta_factor = 0.8
            pi_star = pi_k.min()
            t = float(self._iteration)
            if self._stage == 0:
                gamma = self.gamma0 * min(pi_star, t ** (-beta_factor))                         # :637
            else:
                gamma = self.gamma0 * min(pi_star, (t - self._t0 + self._t0 ** beta_factor) ** (-1))   # :639
            if self.weight_update_method == 'optimal':
                self._logZ[state_index] += gamma * np.exp(-log_pi_k[state_index])              # :649-652
            else:
                log_P_k = replicas_log_P_k[replica_index, :]
                self._logZ[:] += gamma * np.exp(log_P_k - log_pi_k)                            # :659-664
        if self._stage == 1:
            self._logZ[:] -= self._logZ[0]                                                     # :669-670
        self._gamma = gamma

    def _update_log_weights(self):
        """sams.py:683-691."""
        self.log_weights = self.log_target_probabilities[:] - self._logZ[:]
