"""CPU model of a pairwise-rendezvous swap-all (a lead for the next round, DESIGN.md 7d "Leads"; NOT a device kernel).

The reference's swap-all (replicaexchange.py:294-349) is a sequential loop over R^3 attempts (i, j, u); attempt t reads the labels
the attempts before it left.  Only attempts that share a replica are ordered: the model gives every replica one agent that walks
its own chain of attempts (built from the label-independent draws, which a whole-chip kernel can do) and lets the two agents of
an attempt meet:

    head[r]           the attempt replica r waits at (its chain position)
    attempt (i, j)    the LOWER-indexed agent decides it, once the higher one's head points at the same attempt: both replicas have
                      then finished everything earlier that involves them, so their labels are the sequential loop's
    mailbox[h]        the deciding agent leaves the partner's new label there and moves on; the partner picks it up and moves on
    i == j            a solo entry in one chain (the reference counts it as a proposed and accepted swap of a state with itself)

Agents are stepped in ARBITRARY order (``schedule``: 'random' / 'round-robin' / 'reverse'): the result must not depend on it.  The
model returns the labels, the count matrices and the number of synchronous rounds a lock-step execution needs (every agent tries
once per round) -- the dependent depth a device kernel would pay per LDS round trip.

usage: python tools/experiments/mix_rendezvous_model.py [R] [n_attempts]    (prints rounds and decided attempts per round)
"""
import sys

import numpy as np


def build_chains(R, ii, jj):
    """Per-replica lists of attempt numbers in order (label-independent)."""
    chains = [[] for _ in range(R)]
    for t, (i, j) in enumerate(zip(ii, jj)):
        chains[i].append(t)
        if j != i:
            chains[j].append(t)
    return chains


def swap_all_rendezvous(u_kl, labels, ii, jj, uu, accept, schedule='random', seed=0):
    """``accept(log_p, u)`` is the reference's Metropolis test (replicaexchange.py:343).  Returns (labels, n_accepted, n_proposed,
    rounds)."""
    R, K = u_kl.shape
    lab = np.array(labels, dtype=np.int64)
    chains = build_chains(R, ii, jj)
    pos = np.zeros(R, dtype=np.int64)                       # next chain entry of every agent
    mailbox = [None] * R                                    # (attempt, new label) left for a passive agent
    n_acc = np.zeros((K, K), dtype=np.int64)
    n_prop = np.zeros((K, K), dtype=np.int64)
    rng = np.random.default_rng(seed)
    remaining = sum(len(c) for c in chains)
    rounds = 0
    order = np.arange(R)
    while remaining:
        rounds += 1
        if schedule == 'random':
            order = rng.permutation(R)
        elif schedule == 'reverse':
            order = np.arange(R)[::-1]
        progressed = False
        # a lock-step round: what an agent sees of the others is their state at the START of the round
        head = np.array([chains[r][pos[r]] if pos[r] < len(chains[r]) else -1 for r in range(R)])
        lab0 = lab.copy()
        box0 = list(mailbox)
        for r in order:
            if pos[r] >= len(chains[r]):
                continue
            t = chains[r][pos[r]]
            i, j = int(ii[t]), int(jj[t])
            if i == j:                                                              # solo entry
                s = lab[r]
                n_prop[s, s] += 2; n_acc[s, s] += 2                                 # :339-340, 348-349 with si == sj (log_p = 0)
                pos[r] += 1; remaining -= 1; progressed = True
                continue
            p = j if r == i else i
            if r < p:                                                               # active side
                if head[p] == t and box0[p] is None:
                    si, sj = (lab0[i], lab0[j])
                    log_p = -(u_kl[i, sj] + u_kl[j, si]) + u_kl[i, si] + u_kl[j, sj]   # :332-336, the oracle's association
                    n_prop[si, sj] += 1; n_prop[sj, si] += 1
                    new_r, new_p = lab0[r], lab0[p]
                    if accept(log_p, uu[t]):
                        n_acc[si, sj] += 1; n_acc[sj, si] += 1
                        new_r, new_p = lab0[p], lab0[r]
                    lab[r] = new_r
                    mailbox[p] = (t, new_p)
                    pos[r] += 1; remaining -= 1; progressed = True
            else:                                                                   # passive side
                if box0[r] is not None and box0[r][0] == t:
                    lab[r] = box0[r][1]
                    mailbox[r] = None
                    pos[r] += 1; remaining -= 1; progressed = True
        assert progressed, 'deadlock'
    return lab, n_acc, n_prop, rounds


if __name__ == '__main__':
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40 * R * R
    rng = np.random.default_rng(1)
    beta = 1.0 / (0.0083144626 * np.geomspace(300.0, 600.0, R))
    U = -2.4e4 + 150.0 * rng.normal(size=R)
    u_kl = np.outer(U, beta)
    ii, jj, uu = rng.integers(0, R, n), rng.integers(0, R, n), rng.random(n)
    lab, na, npr, rounds = swap_all_rendezvous(u_kl, np.arange(R), ii, jj, uu, lambda lp, u: lp >= 0.0 or u < np.exp(lp))
    chain_entries = 2 * n - int((ii == jj).sum())
    print('R %d attempts %d: %d lock-step rounds, %.1f attempts decided per round, acceptance %.3f; R^3 = %d attempts would take %.0f rounds'
          % (R, n, rounds, n / rounds, na.sum() / max(1, npr.sum()), R ** 3, R ** 3 * rounds / n))
