"""Parity against vectors produced by the REFERENCE'S OWN functions (tests/golden/reference_mix.json, written by
tests/golden/make_golden_from_reference.py: the bodies of replicaexchange.py:294-406, sams.py:395-691 and
multistatesampler.py:1263-1281 executed from /root/reference with only np.random injected).

CPU part: the C oracle (oracle/mix_oracle.c) and the host logZ recursion (openmmtools_amd/multistate/sams.py) must
reproduce them.  GPU part (-m gpu): the HIP kernels through the C ABI must reproduce them bit for bit (labels, count
matrices) / to 1e-12 (log P), and the host recursion fed by the device kernel must follow the reference's trajectory.
"""
import json
import os
import numpy as np
import pytest
import oracle

_PATH = os.path.join(os.path.dirname(__file__), 'golden', 'reference_mix.json')
_GOLD = json.load(open(_PATH))
MIX = _GOLD['mix']
SAMS = _GOLD['sams']


def _id(c):
    return '%s-%s-R%d-it%d' % (c['scheme'], c.get('entry', ''), c['R'], c['iteration'])


def _sid(c):
    return 'sams-R%d-K%d-%s-%s-%s' % (c['R'], c['K'], c['update_stages'], c['flatness_criteria'], c['weight_update_method'])


# ---------------------------------------------------------------------------------------------- C oracle (CPU)
@pytest.mark.parametrize('c', MIX, ids=_id)
def test_c_oracle_reproduces_reference_mixing(c):
    lab, nacc, nprop, _ = oracle.mix(c['scheme'], c['seed'], c['iteration'], np.array(c['u_kl']),
                                     np.array(c['labels_in'], dtype=np.int64))
    assert lab.tolist() == c['labels_out']
    assert nacc.tolist() == c['n_accepted'] and nprop.tolist() == c['n_proposed']


@pytest.mark.parametrize('c', SAMS, ids=_sid)
def test_c_oracle_reproduces_reference_sams(c):
    """oracle_sams_global_jump vs sams.py:477-501 and oracle_sams_update_logZ vs :606-681, frame by frame."""
    import ctypes as C
    lib = oracle.mix_lib()
    K, R = c['K'], c['R']
    log_pi = np.array(c['log_target_probabilities'])
    logZ = np.zeros(K)
    for f in c['frames']:
        lab, nacc, nprop, logP = oracle.mix('sams-global-jump', c['seed'], f['iteration'], np.array(f['u_kl']),
                                            np.array(f['labels_in'], dtype=np.int64), log_weights=np.array(f['log_weights_in']))
        assert lab.tolist() == f['labels_out']
        assert nacc.tolist() == f['n_accepted'] and nprop.tolist() == f['n_proposed']
        assert np.allclose(logP, np.array(f['log_P']), rtol=0, atol=1e-12)
        if f['iteration'] > 0:                                     # sams.py:428
            # stage / t0 are the caller's business in the C oracle: take them from the reference's frame
            gamma = C.c_double(0.0)
            labels = np.ascontiguousarray(lab, dtype=np.int64)
            lp = np.ascontiguousarray(logP)
            lib.oracle_sams_update_logZ(R, K, labels.ctypes.data_as(C.POINTER(C.c_int64)),
                                        lp.ctypes.data_as(C.POINTER(C.c_double)),
                                        log_pi.ctypes.data_as(C.POINTER(C.c_double)), c['gamma0'], f['iteration'],
                                        f['stage'], f['t0'], 1 if c['weight_update_method'] == 'optimal' else 0,
                                        logZ.ctypes.data_as(C.POINTER(C.c_double)), C.byref(gamma))
            assert np.allclose(logZ, np.array(f['logZ']), rtol=1e-13, atol=1e-13)
            assert abs(gamma.value - f['gamma']) <= 1e-15 * abs(f['gamma'])
        logZ = np.array(f['logZ'])                                  # do not let round-off accumulate across frames


# ---------------------------------------------------------------------------------------------- host sams.py (CPU)
class _ReplayEngine:
    """Feeds SAMSSampler._mix_replicas: mixing by a callable (C oracle on CPU, the HIP kernel on the GPU)."""
    is_device = False

    def __init__(self, mix_fn):
        self.mix_fn = mix_fn
        self.u = None

    def mix(self, scheme, it, labels, R=None, K=None, ld=None, log_weights=None, **kw):
        return self.mix_fn(scheme, it, self.u, labels, log_weights)


def _make_sams(c, engine):
    from openmmtools_amd.multistate import SAMSSampler
    K, R = c['K'], c['R']
    s = SAMSSampler(number_of_iterations=10 ** 6, engine=engine, seed=c['seed'], update_stages=c['update_stages'],
                    flatness_criteria=c['flatness_criteria'], flatness_threshold=c['flatness_threshold'],
                    weight_update_method=c['weight_update_method'], gamma0=c['gamma0'],
                    log_target_probabilities=np.array(c['log_target_probabilities']))
    # what _pre_write_create would have set up (sams.py:301-372), without a System
    s._thermodynamic_states = [None] * K
    s._sampler_states = [None] * R
    s._K_total = K
    s._n_accepted_matrix = np.zeros((K, K), np.int64)
    s._n_proposed_matrix = np.zeros((K, K), np.int64)
    s._neighborhoods = np.ones((R, K), np.int8)
    s._initialize_stage()
    s.log_target_probabilities = np.array(c['log_target_probabilities'])
    s._logZ = np.zeros(K)
    s._update_log_weights()
    s._cached_state_histogram = np.zeros(K, dtype=int)
    s._replica_thermodynamic_states = np.array(c['frames'][0]['labels_in'], dtype=np.int64)
    s._report_iteration()          # create() reports iteration 0: SAMS counts the initial states (sams.py:381-393)
    return s


def _follow_reference_trajectory(c, mix_fn):
    eng = _ReplayEngine(mix_fn)
    s = _make_sams(c, eng)
    for f in c['frames']:
        eng.u = np.array(f['u_kl'])
        s._iteration = f['iteration']
        assert s._replica_thermodynamic_states.tolist() == f['labels_in']
        assert np.allclose(s.log_weights, f['log_weights_in'], rtol=1e-12, atol=1e-12)
        labels = s._mix_replicas()
        s._replica_thermodynamic_states = np.asarray(labels)
        assert np.asarray(labels).tolist() == f['labels_out']
        assert s._n_accepted_matrix.tolist() == f['n_accepted'] and s._n_proposed_matrix.tolist() == f['n_proposed']
        assert (s._stage, s._t0) == (f['stage'], f['t0']), 'stage/t0 at iteration %d' % f['iteration']
        assert np.allclose(s._logZ, f['logZ'], rtol=1e-11, atol=1e-11)
        assert np.allclose(s.log_weights, f['log_weights_out'], rtol=1e-11, atol=1e-11)
        if f['gamma'] is not None:
            assert abs(s._gamma - f['gamma']) <= 1e-14 * abs(f['gamma'])
        assert s._cached_state_histogram.tolist() == f['histogram_before_report']
        # the reporter step (sams.py:381-393) counts the visited states
        states, counts = np.unique(s._replica_thermodynamic_states, return_counts=True)
        s._cached_state_histogram[states] += counts


@pytest.mark.parametrize('c', SAMS, ids=_sid)
def test_host_sams_recursion_follows_reference(c):
    def mix_fn(scheme, it, u, labels, log_weights):
        return oracle.mix(scheme, c['seed'], it, u, labels, log_weights=log_weights)
    _follow_reference_trajectory(c, mix_fn)


# ---------------------------------------------------------------------------------------------- HIP kernels (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize('c', MIX, ids=_id)
def test_hip_mixing_reproduces_reference(c, hip_engine_factory):
    eng = hip_engine_factory()
    eng.seed(c['seed'])
    lab, nacc, nprop, _ = eng.mix_host(c['scheme'], c['iteration'], np.array(c['u_kl']), np.array(c['labels_in'], dtype=np.int64))
    assert lab.tolist() == c['labels_out']
    assert nacc.tolist() == c['n_accepted'] and nprop.tolist() == c['n_proposed']


@pytest.mark.gpu
@pytest.mark.parametrize('c', SAMS, ids=_sid)
def test_hip_sams_follows_reference_trajectory(c, hip_engine_factory):
    """sams_global_jump_kernel (csrc/mix.hip) + host recursion == the reference's _mix_replicas, iteration by iteration
    (config 5's shape R = 16, K = 128 is one of the cases)."""
    eng = hip_engine_factory()
    eng.seed(c['seed'])
    logPs = {}

    def mix_fn(scheme, it, u, labels, log_weights):
        out = eng.mix_host(scheme, it, u, labels, log_weights=log_weights)
        logPs[it] = out[3]
        return out
    _follow_reference_trajectory(c, mix_fn)
    for f in c['frames']:
        assert np.allclose(logPs[f['iteration']], np.array(f['log_P']), rtol=0, atol=1e-12)
