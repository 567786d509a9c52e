"""Generates tests/golden/mix_reference_arith.json: the swap-all mixing of openmmtools/multistate/replicaexchange.py
(:294-349 loop, :382-406 per-attempt arithmetic) evaluated by a pure-Python line-by-line transcription (math.exp, Python
ints) on the (i, j, u) sequence of this repository's Philox stream spec.  The reference module itself cannot be imported
here (it needs numba / openmm / mpiplus), so the transcription is the pinned golden vector; the C oracle and the
HIP kernel must both reproduce it bit for bit (tests/test_oracle_mix.py, tests/test_mix_parity.py).

usage: python tests/golden/make_golden_mix.py   (test infrastructure: the only place outside tests/ proper, smoke() and
bench.py's cpu_baseline leg that touches oracle/ is this fixture generator, which lives under tests/)
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import md_oracle as mo


def attempt_swap(u, labels, nacc, nprop, i, j, r):
    """replicaexchange.py:382-406."""
    si, sj = labels[i], labels[j]
    energy_ij, energy_ji = u[i][sj], u[j][si]
    energy_ii, energy_jj = u[i][si], u[j][sj]
    log_p_accept = - (energy_ij + energy_ji) + energy_ii + energy_jj
    nprop[si][sj] += 1
    nprop[sj][si] += 1
    if log_p_accept >= 0.0 or r < math.exp(log_p_accept):
        labels[i], labels[j] = sj, si
        nacc[si][sj] += 1
        nacc[sj][si] += 1


def case(R, seed, iteration, scale):
    rng = np.random.default_rng(1000 + R)
    u = (np.outer(rng.normal(scale=scale, size=R), np.linspace(0.5, 1.5, R)) + rng.normal(scale=0.5, size=(R, R))).tolist()
    labels = [int(x) for x in rng.permutation(R)]
    lab = list(labels)
    nacc = [[0] * R for _ in range(R)]
    nprop = [[0] * R for _ in range(R)]
    for k in range(R ** 3):                                             # :269 nswap_attempts = n_replicas ** 3
        w = [int(x) for x in mo.draw(seed, mo.STREAM_SWAP_ALL, k & 0xFFFFFFFF, k >> 32, iteration)]
        i, j = (w[0] * R) >> 32, (w[1] * R) >> 32                       # :324-325 randint(n_replicas)
        r = ((w[2] << 21) | (w[3] >> 11)) / 9007199254740992.0
        attempt_swap(u, lab, nacc, nprop, i, j, r)
    return dict(R=R, seed=seed, iteration=iteration, u_kl=u, labels_in=labels, labels_out=lab, n_accepted=nacc, n_proposed=nprop)


if __name__ == '__main__':
    cases = [case(4, 0xC0FFEE, 0, 3.0), case(9, 12345, 7, 2.0), case(24, 0xC0FFEE, 3, 3.0)]
    out = os.path.join(ROOT, 'tests', 'golden', 'mix_reference_arith.json')
    with open(out, 'w') as fh:
        json.dump(dict(generator='tests/golden/make_golden_mix.py', cases=cases), fh)
    print('wrote', out, os.path.getsize(out), 'bytes')
