"""Golden vectors produced BY THE REFERENCE'S OWN CODE for the integer / label path.

The reference package cannot be imported here (openmm / numba / mpiplus are absent), but the functions of
the mixing path are plain Python + numpy once their decorators are gone.  This script reads the reference source files
where they lie under /root/reference, takes the function definitions out of the module syntax tree *unchanged*
(only the decorator lists are dropped: ``@njit``, ``@staticmethod``, ``@mpiplus.on_single_node`` ...), compiles them and
runs them on a stand-in ``self`` that owns the same arrays the sampler owns.  The only thing injected is the random
number source: the name ``np`` seen by the reference functions is a proxy whose ``random`` serves this repository's
Philox stream spec with numpy's own algorithms on top of the raw uniforms

    np.random.randint(n)         -> mulhi(w, n)                 (w0 = replica i, w1 = replica j of swap attempt k)
    np.random.rand()             -> 53-bit uniform of (w2, w3)
    np.random.choice(a, p=p)     -> numpy's legacy algorithm: cdf = cumsum(p); cdf /= cdf[-1];
                                    a[searchsorted(cdf, u, side='right')]

Functions executed (file:line of the reference):
    replicaexchange.py:294-349  _mix_all_replicas_numba      (numba-free: the body is plain Python)
    replicaexchange.py:351-364  _mix_all_replicas
    replicaexchange.py:366-380  _mix_neighboring_replicas
    replicaexchange.py:382-406  _attempt_swap
    sams.py:395-437             SAMSSampler._mix_replicas
    sams.py:477-501             _global_jump
    sams.py:564-604             _update_stage
    sams.py:606-681             _update_logZ_estimates
    sams.py:683-691             _update_log_weights
    multistatesampler.py:1263-1281  _neighborhood

Output: tests/golden/reference_mix.json (small; committed).  /root/reference does not exist on the GPU box, so the
tests read only the JSON.   usage:  python tests/golden/make_golden_from_reference.py
"""
import ast
import contextlib
import json
import logging
import math
import os
import sys
import warnings

import numpy as real_np
from scipy.special import logsumexp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle                                    # only the Philox block function (pinned by the Random123 KATs)

REF = '/root/reference/openmmtools/multistate'


# ------------------------------------------------------------------------------------------------------------------
# the injected random source
# ------------------------------------------------------------------------------------------------------------------
def u53(w2, w3):
    return ((int(w2) << 21) | (int(w3) >> 11)) / 9007199254740992.0


class PhiloxRandom:
    """Stands in for ``np.random`` inside the reference functions.  ``key`` says which stream/counter the next
    draws come from: ('swap-all',) advances an attempt index on every first randint of a pair; ('pair', state) and
    ('replica', r) are set by the callers below just before the reference function that draws."""

    def __init__(self, seed, iteration):
        self.seed, self.iteration = seed, iteration
        self.mode = None
        self.k = -1                # swap-all attempt index
        self.phase = 0
        self.w = None
        self.log = []              # every number handed out, for the fixture's provenance

    def _draw(self, stream, a, b):
        return [int(x) for x in oracle.draw(self.seed, stream, a, b, self.iteration)]

    def randint(self, n):
        if self.mode == 'swap-all':
            if self.phase == 0:
                self.k += 1
                self.w = self._draw(1, self.k & 0xFFFFFFFF, self.k >> 32)
                self.phase = 1
                return (self.w[0] * n) >> 32
            self.phase = 0
            return (self.w[1] * n) >> 32
        if self.mode == 'neighbors-offset':
            assert n == 2
            self.w = self._draw(2, 0, 0)
            return self.w[0] & 1
        raise AssertionError('randint in mode %r' % self.mode)

    def rand(self):
        if self.mode == 'swap-all':
            return u53(self.w[2], self.w[3])
        if self.mode == 'pair':
            return u53(self.w[2], self.w[3])
        raise AssertionError('rand in mode %r' % self.mode)

    def set_pair(self, state):
        self.mode = 'pair'
        self.w = self._draw(2, 1 + int(state), 0)

    def set_replica(self, r):
        self.mode = 'replica'
        self.w = self._draw(3, int(r), 0)

    def choice(self, a, p=None):
        assert self.mode == 'replica'
        # numpy/random/mtrand.pyx RandomState.choice with p given, size None, replace True
        p = real_np.asarray(p, dtype=real_np.float64)
        cdf = p.cumsum()
        cdf /= cdf[-1]
        u = u53(self.w[2], self.w[3])
        idx = int(cdf.searchsorted(u, side='right'))
        return a[idx]


class NumpyProxy:
    """``np`` as the reference functions see it: everything is real numpy except ``random``."""

    def __init__(self, rnd):
        self.random = rnd

    def __getattr__(self, name):
        return getattr(real_np, name)


class _Utils:
    @staticmethod
    @contextlib.contextmanager
    def time_it(name):
        yield


# ------------------------------------------------------------------------------------------------------------------
# taking the functions out of the reference source
# ------------------------------------------------------------------------------------------------------------------
def extract(path, class_name, names, namespace):
    """Compile the named methods of ``class_name`` from the file at ``path`` with empty decorator lists."""
    with open(path) as fh:
        tree = ast.parse(fh.read(), path)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == class_name][0]
    out = {}
    for node in cls.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            mod = ast.Module(body=[node], type_ignores=[])
            code = compile(mod, path, 'exec')          # keeps the reference's file name and line numbers
            ns = dict(namespace)
            exec(code, ns)
            out[node.name] = ns[node.name]
            out[node.name].__ref_lines__ = (node.lineno, node.end_lineno)
    missing = set(names) - set(out)
    assert not missing, missing
    return out


class _Reporter:
    def __init__(self):
        self.records = []

    def write_online_analysis_data(self, iteration, **kw):
        self.records.append((int(iteration), {k: (real_np.array(v).tolist()) for k, v in kw.items()}))


def make_reference_classes(rnd):
    ns = dict(np=NumpyProxy(rnd), math=math, logsumexp=logsumexp, logger=logging.getLogger('golden'), utils=_Utils)
    rex = extract(os.path.join(REF, 'replicaexchange.py'), 'ReplicaExchangeSampler',
                  ['_mix_all_replicas_numba', '_mix_all_replicas', '_mix_neighboring_replicas', '_attempt_swap'], ns)
    sams = extract(os.path.join(REF, 'sams.py'), 'SAMSSampler',
                   ['_mix_replicas', '_global_jump', '_update_stage', '_update_logZ_estimates', '_update_log_weights'], ns)
    mss = extract(os.path.join(REF, 'multistatesampler.py'), 'MultiStateSampler', ['_neighborhood'], ns)

    ref_attempt_swap = rex['_attempt_swap']

    class RefRex:
        """Owns what ReplicaExchangeSampler owns for mixing (replicaexchange.py:261-262, multistatesampler.py:892-895)."""

        def __init__(self, u, labels):
            R, K = u.shape
            self.n_replicas, self.n_states = R, K
            self._energy_thermodynamic_states = real_np.array(u, dtype=real_np.float64)
            self._replica_thermodynamic_states = real_np.array(labels, dtype=real_np.int64)
            self._n_accepted_matrix = real_np.zeros((K, K), real_np.int64)
            self._n_proposed_matrix = real_np.zeros((K, K), real_np.int64)

        _mix_all_replicas_numba = staticmethod(rex['_mix_all_replicas_numba'])
        _mix_all_replicas = rex['_mix_all_replicas']
        _mix_neighboring_replicas = rex['_mix_neighboring_replicas']

        def _attempt_swap(self, replica_i, replica_j):
            if rnd.mode in ('neighbors-offset', 'pair'):
                # the pair of neighbouring states (s, s+1) keys its uniform by s (stream 2, a = 1 + s)
                s = int(real_np.asarray(self._replica_thermodynamic_states[replica_i]).ravel()[0])
                rnd.set_pair(s)
            return ref_attempt_swap(self, replica_i, replica_j)

    ref_global_jump = sams['_global_jump']

    class RefSAMS(RefRex):
        def __init__(self, u, labels, log_target, gamma0, update_stages, flatness_criteria, flatness_threshold,
                     weight_update_method):
            super().__init__(u, labels)
            K = self.n_states
            self.locality = None                                              # sams.py:338-339
            self.state_update_scheme = 'global-jump'
            self.update_stages, self.flatness_criteria = update_stages, flatness_criteria
            self.flatness_threshold, self.weight_update_method, self.gamma0 = flatness_threshold, weight_update_method, gamma0
            self.log_target_probabilities = real_np.array(log_target, real_np.float64)
            self._logZ = real_np.zeros(K)
            self._t0 = 0                                                      # sams.py:291-296
            self._stage = 1 if update_stages == 'one-stage' else 0
            self._iteration = 0
            self._neighborhoods = real_np.ones((self.n_replicas, K), real_np.int8)
            self._cached_state_histogram = real_np.zeros(K, dtype=int)
            self._reporter = _Reporter()
            self._update_log_weights()

        _state_histogram = property(lambda self: self._cached_state_histogram)     # sams.py:540-552 (cached branch)
        _neighborhood = mss['_neighborhood']
        _mix_replicas = sams['_mix_replicas']
        _update_stage = sams['_update_stage']
        _update_logZ_estimates = sams['_update_logZ_estimates']
        _update_log_weights = sams['_update_log_weights']

        def _global_jump(self, replicas_log_P_k):
            # the reference loops over replicas inside; key each replica's categorical draw by its index
            # iterating enumerate(labels) is the only place the replica index is known: wrap the array
            labels = self._replica_thermodynamic_states

            class KeyedLabels(real_np.ndarray):
                def __iter__(inner):
                    for r, s in enumerate(real_np.asarray(inner).tolist()):
                        rnd.set_replica(r)
                        yield s
            self._replica_thermodynamic_states = labels.view(KeyedLabels)
            try:
                ref_global_jump(self, replicas_log_P_k)
            finally:
                self._replica_thermodynamic_states = real_np.asarray(self._replica_thermodynamic_states).view(real_np.ndarray)

        def report(self):
            """sams.py:381-393: the histogram is counted when an iteration is reported."""
            states, counts = real_np.unique(self._replica_thermodynamic_states, return_counts=True)
            self._cached_state_histogram[states] += counts

    return RefRex, RefSAMS


# ------------------------------------------------------------------------------------------------------------------
# cases
# ------------------------------------------------------------------------------------------------------------------
def pt_like_u(R, K, scale, rng):
    return real_np.outer(rng.normal(scale=scale, size=R), real_np.linspace(0.5, 1.5, K)) + rng.normal(scale=0.5, size=(R, K))


def swap_all_case(R, seed, iteration, scale, entry):
    rng = real_np.random.default_rng(1000 + R)
    u = pt_like_u(R, R, scale, rng)
    labels = rng.permutation(R).astype(real_np.int64)
    rnd = PhiloxRandom(seed, iteration)
    RefRex, _ = make_reference_classes(rnd)
    s = RefRex(u, labels)
    rnd.mode = 'swap-all'
    if entry == 'numba':                                              # replicaexchange.py:271-276 call shape
        s._mix_all_replicas_numba(R ** 3, R, s._replica_thermodynamic_states, s._energy_thermodynamic_states,
                                  s._n_accepted_matrix, s._n_proposed_matrix)
    else:
        s._mix_all_replicas(R ** 3)                                   # :280
    return dict(scheme='swap-all', entry=entry, R=R, K=R, seed=seed, iteration=iteration, u_kl=u.tolist(),
                labels_in=labels.tolist(), labels_out=s._replica_thermodynamic_states.tolist(),
                n_accepted=s._n_accepted_matrix.tolist(), n_proposed=s._n_proposed_matrix.tolist())


def neighbors_case(R, seed, iteration, scale):
    rng = real_np.random.default_rng(2000 + R)
    u = pt_like_u(R, R, scale, rng)
    labels = rng.permutation(R).astype(real_np.int64)
    rnd = PhiloxRandom(seed, iteration)
    RefRex, _ = make_reference_classes(rnd)
    s = RefRex(u, labels)
    rnd.mode = 'neighbors-offset'
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', DeprecationWarning)           # math.exp on the 1-element arrays np.where gives
        s._mix_neighboring_replicas()
    return dict(scheme='swap-neighbors', R=R, K=R, seed=seed, iteration=iteration, u_kl=u.tolist(),
                labels_in=labels.tolist(), labels_out=s._replica_thermodynamic_states.tolist(),
                n_accepted=s._n_accepted_matrix.tolist(), n_proposed=s._n_proposed_matrix.tolist())


def sams_trajectory(R, K, seed, n_iterations, update_stages, flatness_criteria, flatness_threshold, weight_update_method,
                    gamma0=1.0, spread=2.0):
    """n_iterations calls of the reference's SAMSSampler._mix_replicas (jump + logZ update + weights) on synthetic
    energies that change every iteration; the histogram is counted after each one as the reporter step does."""
    rng = real_np.random.default_rng(3000 + R * 131 + K)
    log_target = real_np.zeros(K) - real_np.log(K)
    f_true = real_np.linspace(0.0, spread, K) ** 1.5                  # the free energies the weights should learn
    labels = rng.integers(0, K, size=R).astype(real_np.int64)
    rnd = PhiloxRandom(seed, 0)
    _, RefSAMS = make_reference_classes(rnd)
    s = RefSAMS(real_np.zeros((R, K)), labels, log_target, gamma0, update_stages, flatness_criteria, flatness_threshold,
                weight_update_method)
    s.report()              # create() reports iteration 0 (multistatesampler.py:588-609): the initial states are counted
    frames = []
    for it in range(n_iterations):
        u = f_true[None, :] + rng.normal(scale=1.0, size=(R, K))      # -ln of unnormalised densities + noise
        s._energy_thermodynamic_states = u
        s._iteration = it
        rnd.iteration = it
        labels_in = s._replica_thermodynamic_states.copy()
        weights_in = s.log_weights.copy()
        s._reporter.records.clear()
        out = s._mix_replicas()
        rec = dict(s._reporter.records and [(k, v) for _, d in s._reporter.records for k, v in d.items()] or [])
        # log P of the jump: _mix_replicas keeps it local, recompute it the way :486-489 does for the fixture
        logP = -u + weights_in[None, :]
        logP = logP - logsumexp(logP, axis=1)[:, None]
        frames.append(dict(iteration=it, u_kl=u.tolist(), labels_in=labels_in.tolist(), log_weights_in=weights_in.tolist(),
                           labels_out=real_np.asarray(out).tolist(), log_P=logP.tolist(),
                           n_accepted=s._n_accepted_matrix.tolist(), n_proposed=s._n_proposed_matrix.tolist(),
                           logZ=s._logZ.tolist(), log_weights_out=s.log_weights.tolist(), stage=int(s._stage), t0=int(s._t0),
                           gamma=rec.get('gamma'), stored_log_weights=rec.get('log_weights'),
                           histogram_before_report=s._cached_state_histogram.tolist()))
        s.report()
    return dict(scheme='sams-global-jump', R=R, K=K, seed=seed, gamma0=gamma0, update_stages=update_stages,
                flatness_criteria=flatness_criteria, flatness_threshold=flatness_threshold,
                weight_update_method=weight_update_method, log_target_probabilities=log_target.tolist(), frames=frames)


def main():
    real_np.seterr(all='raise', under='ignore')
    cases = []
    for R, seed, it, scale, entry in [(4, 0xC0FFEE, 0, 3.0, 'numba'), (4, 0xC0FFEE, 0, 3.0, 'python'),
                                      (9, 12345, 7, 2.0, 'numba'), (9, 12345, 7, 2.0, 'python'),
                                      (24, 0xC0FFEE, 3, 3.0, 'numba'), (64, 0xC0FFEE, 11, 4.0, 'numba')]:
        c = swap_all_case(R, seed, it, scale, entry)
        cases.append(c)
        print('swap-all', entry, 'R', R, 'accepted', int(real_np.sum(c['n_accepted'])) // 2, 'of', R ** 3)
    for R, seed, it in [(4, 0xC0FFEE, 0), (9, 12345, 7), (9, 12345, 8), (24, 0xC0FFEE, 3), (64, 99, 5), (65, 99, 6)]:
        c = neighbors_case(R, seed, it, 2.0)
        cases.append(c)
        print('swap-neighbors R', R, 'accepted', int(real_np.sum(c['n_accepted'])) // 2)
    sams = [sams_trajectory(5, 7, 0xC0FFEE, 40, 'two-stage', 'logZ-flatness', 0.2, 'rao-blackwellized'),
            sams_trajectory(5, 7, 0xC0FFEE, 25, 'two-stage', 'minimum-visits', 0.2, 'optimal'),
            sams_trajectory(3, 6, 4242, 60, 'two-stage', 'histogram-flatness', 0.9, 'rao-blackwellized', spread=0.5),
            sams_trajectory(4, 5, 77, 12, 'one-stage', 'logZ-flatness', 0.2, 'rao-blackwellized'),
            sams_trajectory(16, 128, 0xC0FFEE, 6, 'two-stage', 'logZ-flatness', 0.2, 'rao-blackwellized', spread=3.0)]
    for c in sams:
        print('sams R %d K %d %s/%s: final stage %d t0 %d' % (c['R'], c['K'], c['flatness_criteria'], c['weight_update_method'],
                                                             c['frames'][-1]['stage'], c['frames'][-1]['t0']))
    out = os.path.join(ROOT, 'tests', 'golden', 'reference_mix.json')
    with open(out, 'w') as fh:
        json.dump(dict(generator='tests/golden/make_golden_from_reference.py',
                       provenance='function bodies executed from /root/reference/openmmtools/multistate/'
                                  '{replicaexchange,sams,multistatesampler}.py (decorators stripped, np.random injected)',
                       mix=cases, sams=sams), fh, separators=(',', ':'))
    print('wrote', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
