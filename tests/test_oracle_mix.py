"""Pins the CPU oracle for the mixing path (CPU-only tests).

  * Philox4x32-10 against the published Random123 known-answer vectors (kat_vectors)
  * the per-attempt arithmetic against a pure-Python transcription of
    ReplicaExchangeSampler._attempt_swap (openmmtools/multistate/replicaexchange.py:382-406)
    driven by the same (i, j, u) sequence
  * the reference's own distributional test: chi-square uniformity of visited labels with
    u_kl = 0 (openmmtools/tests/test_mixing.py:11-46, 76-92)
"""
import math
import numpy as np
import scipy.stats
import pytest
import oracle
from oracle import md_oracle as mo

KATS = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize('ctr,key,expect', KATS)
def test_philox_known_answers(ctr, key, expect):
    assert [int(x) for x in oracle.philox(ctr, key)] == expect
    assert [int(x) for x in mo.philox4x32_10(*ctr, *key)] == expect


def test_stream_layout_c_vs_numpy():
    for (seed, stream, a, b, t) in [(0xC0FFEE, 1, 5, 0, 7), (2**63 + 12345, 5, 2268, 23, 2**33 + 17), (1, 3, 0, 0, 0)]:
        c = oracle.draw(seed, stream, a, b, t)
        n = mo.draw(seed, stream, a, b, t)
        assert [int(x) for x in c] == [int(x) for x in n]


def test_exp_det_accuracy():
    xs = -np.concatenate([np.linspace(0, 50, 2001), np.logspace(-12, 2.8, 500)])
    rel = [abs(oracle.exp_det(x) / math.exp(x) - 1.0) for x in xs if x > -700]
    assert max(rel) < 4e-16
    assert oracle.exp_det(-800.0) == 0.0 and oracle.exp_det(0.0) == 1.0


def _attempt_swap_python(u, labels, nacc, nprop, i, j, r):
    """Line-by-line transcription of replicaexchange.py:382-406 (math.exp instead of exp_det)."""
    si, sj = labels[i], labels[j]
    energy_ij, energy_ji = u[i, sj], u[j, si]
    energy_ii, energy_jj = u[i, si], u[j, sj]
    log_p_accept = - (energy_ij + energy_ji) + energy_ii + energy_jj
    nprop[si, sj] += 1
    nprop[sj, si] += 1
    if log_p_accept >= 0.0 or r < math.exp(log_p_accept):
        labels[i], labels[j] = sj, si
        nacc[si, sj] += 1
        nacc[sj, si] += 1


@pytest.mark.parametrize('R', [4, 16, 24])
def test_attempt_arithmetic_matches_reference_transcription(R):
    rng = np.random.default_rng(R)
    u = rng.normal(scale=3.0, size=(R, R))
    n = 4000
    ii, jj, uu = rng.integers(0, R, n), rng.integers(0, R, n), rng.random(n)
    labels = rng.permutation(R).astype(np.int64)
    lab_py, nacc, nprop = labels.copy(), np.zeros((R, R), np.int64), np.zeros((R, R), np.int64)
    for i, j, r in zip(ii, jj, uu):
        _attempt_swap_python(u, lab_py, nacc, nprop, i, j, r)
    lab_c, nacc_c, nprop_c = oracle.mix_sequence(u, labels, ii, jj, uu)
    assert np.array_equal(lab_c, lab_py) and np.array_equal(nacc_c, nacc) and np.array_equal(nprop_c, nprop)


def test_swap_all_stream_matches_python_loop():
    """The Philox-driven loop equals the transcription fed with the same draws."""
    R, seed, it = 12, 0xC0FFEE, 3
    rng = np.random.default_rng(1)
    u = rng.normal(scale=2.0, size=(R, R))
    labels = np.arange(R, dtype=np.int64)
    ii, jj, uu = [], [], []
    for k in range(R ** 3):
        w = [int(x) for x in mo.draw(seed, mo.STREAM_SWAP_ALL, k, 0, it)]
        ii.append((w[0] * R) >> 32); jj.append((w[1] * R) >> 32)
        uu.append(((w[2] << 21) | (w[3] >> 11)) / 2.0 ** 53)
    lab_py, nacc, nprop = labels.copy(), np.zeros((R, R), np.int64), np.zeros((R, R), np.int64)
    for i, j, r in zip(ii, jj, uu):
        _attempt_swap_python(u, lab_py, nacc, nprop, i, j, r)
    lab, a, p, _ = oracle.mix('swap-all', seed, it, u, labels)
    assert np.array_equal(lab, lab_py) and np.array_equal(a, nacc) and np.array_equal(p, nprop)
    assert p.sum() == 2 * R ** 3


def test_uniform_mixing_chi_square():
    """openmmtools/tests/test_mixing.py:76-92 with our stream: u_kl = 0 => every label equally likely."""
    n_states, n_calls = 16, 400
    u = np.zeros((n_states, n_states))
    counts = np.zeros((n_states, n_states))
    labels = np.arange(n_states, dtype=np.int64)
    for call in range(n_calls):
        labels, _, _, _ = oracle.mix('swap-all', 42, call, u, labels)
        counts[np.arange(n_states), labels] += 1
    for r in range(n_states):
        _, p = scipy.stats.chisquare(counts[r])
        assert p > 0.001 / n_states


def test_neighbor_swaps_only_touch_neighbors():
    R = 9
    rng = np.random.default_rng(0)
    u = rng.normal(size=(R, R))
    labels = rng.permutation(R).astype(np.int64)
    for it in range(20):
        new, nacc, nprop, _ = oracle.mix('swap-neighbors', 7, it, u, labels)
        assert sorted(new) == list(range(R))
        prop = np.argwhere(nprop > 0)
        assert all(abs(a - b) == 1 for a, b in prop)
        assert nprop.sum() in (2 * ((R - 1) // 2), 2 * (R // 2))
        labels = new


def test_sams_global_jump_properties():
    R, K = 5, 7
    rng = np.random.default_rng(3)
    u = rng.normal(scale=2.0, size=(R, K))
    logw = rng.normal(size=K)
    labels = rng.integers(0, K, R)
    new, nacc, nprop, logP = oracle.mix('sams-global-jump', 11, 2, u, labels, log_weights=logw)
    from scipy.special import logsumexp
    ref = -u + logw[None, :]
    ref -= logsumexp(ref, axis=1)[:, None]                      # sams.py:486-491
    assert np.allclose(logP, ref, atol=1e-12)
    assert nacc.sum() == R and nprop.sum() == R * K             # sams.py:499-501
    for r in range(R):
        assert nacc[labels[r], new[r]] >= 1
    # draws follow P_k: many iterations, chi-square per replica
    cnt = np.zeros((R, K))
    for it in range(3000):
        nl, _, _, _ = oracle.mix('sams-global-jump', 5, it, u, labels, log_weights=logw)
        cnt[np.arange(R), nl] += 1
    for r in range(R):
        _, p = scipy.stats.chisquare(cnt[r], 3000 * np.exp(ref[r]))
        assert p > 1e-4


def test_golden_mixing_vectors():
    """tests/golden/mix_reference_arith.json (tests/golden/make_golden_mix.py): the pure-Python transcription of
    replicaexchange.py:294-349 / :382-406 on this repository's Philox draws, committed as a fixture."""
    import json, os
    path = os.path.join(os.path.dirname(__file__), 'golden', 'mix_reference_arith.json')
    for c in json.load(open(path))['cases']:
        u = np.array(c['u_kl'])
        lab, nacc, nprop, _ = oracle.mix('swap-all', c['seed'], c['iteration'], u, np.array(c['labels_in'], dtype=np.int64))
        assert lab.tolist() == c['labels_out']
        assert nacc.tolist() == c['n_accepted'] and nprop.tolist() == c['n_proposed']


@pytest.mark.parametrize('schedule', ['random', 'round-robin', 'reverse'])
def test_rendezvous_formulation_of_swap_all_is_the_sequential_loop(schedule):
    """A lead for a device kernel (tools/experiments/mix_rendezvous_model.py): one agent per replica walks its own
    chain of attempts, the lower-indexed agent of a pair decides an attempt once both heads point at it.  Whatever the order the
    agents are stepped in, the labels and the count matrices are those of the sequential oracle, bit for bit (self-swaps and
    repeated pairs included), and the execution never deadlocks."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tools', 'experiments'))
    from mix_rendezvous_model import swap_all_rendezvous
    for R, n, seed in ((5, 400, 1), (12, 3000, 2), (33, 6000, 3)):
        rng = np.random.default_rng(seed)
        u_kl = rng.normal(size=(R, R)) * (0.2 if seed == 1 else 3.0)          # high and low acceptance
        ii, jj, uu = rng.integers(0, R, n).astype(np.int32), rng.integers(0, R, n).astype(np.int32), rng.random(n)
        labels0 = rng.permutation(R).astype(np.int64)
        ref_labels, ref_acc, ref_prop = oracle.mix_sequence(u_kl, labels0, ii, jj, uu)
        accept = lambda log_p, u: log_p >= 0.0 or u < oracle.exp_det(log_p)
        labels, n_acc, n_prop, rounds = swap_all_rendezvous(u_kl, labels0, ii, jj, uu, accept, schedule=schedule, seed=seed)
        assert np.array_equal(labels, ref_labels) and np.array_equal(n_acc, ref_acc) and np.array_equal(n_prop, ref_prop)
        assert 0 < rounds <= 2 * n + R
