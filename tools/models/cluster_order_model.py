"""Round 6: cluster-pair steps of the pair kernel's lists on the headline system under different orders of the molecules (CPU model of the
device's list geometry, as tools/pair_lane_utilisation.py): Z-order / Hilbert curve on n^3 cells, and a recursive bisection of every
64-atom tile into its 8 clusters on top.  Result: Z-order on 7^3 cells (the device until round 6) 27 970 steps per replica, Hilbert on
16^3 cells 23 389, + bisection 22 790.  The Hilbert order was built (forces.hip: hilbert3, profiles/r06_32_hilbert_order.txt); the
bisection was not (2.6 % of the steps).
usage: python tools/models/cluster_order_model.py"""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from openmmtools_amd import testsystems as ts
RC = 1.126
al = ts.AlanineDipeptideExplicit()
x = np.asarray(al.positions, dtype=np.float64)
L = np.diag(al.system.getDefaultPeriodicBoxVectors()).astype(np.float64)
N = x.shape[0]
mol_first = [0] + list(range(22, N, 3)); mol_size = [22] + [3] * ((N - 22) // 3)
def morton(c):
    k = 0
    for b in range(10):
        for a in range(3): k |= ((int(c[a]) >> b) & 1) << (3 * b + a)
    return k
def hilbert3(c, bits):
    # Skilling's transpose-to-hilbert
    X = [int(v) for v in c]; n = 3; M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(n):
            if X[i] & Q: X[0] ^= P
            else:
                t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t
        Q >>= 1
    for i in range(1, n): X[i] ^= X[i - 1]
    t = 0; Q = M
    while Q > 1:
        if X[n - 1] & Q: t ^= Q - 1
        Q >>= 1
    for i in range(n): X[i] ^= t
    k = 0
    for b in range(bits - 1, -1, -1):
        for i in range(n): k = (k << 1) | ((X[i] >> b) & 1)
    return k
def order_atoms(cells, curve):
    keys = []
    for f in mol_first:
        frac = (x[f] / L) % 1.0
        c = np.minimum((frac * cells).astype(int), cells - 1)
        keys.append(morton(c) if curve == 'morton' else hilbert3(c, int(np.ceil(np.log2(cells)))))
    order = np.argsort(np.array(keys), kind='stable')
    return np.concatenate([np.arange(mol_first[m], mol_first[m] + mol_size[m]) for m in order])
def unwrap(p):
    d = p - p[0]; d -= L * np.round(d / L); return p[0] + d
def bisect(idx, pts, size):
    if len(idx) <= size: return [idx]
    q = pts[idx]; ax = np.argmax(q.max(0) - q.min(0))
    o = idx[np.argsort(q[:, ax], kind='stable')]
    half = (len(o) // 2 + size - 1) // size * size if len(o) > size else len(o)
    half = min(half, len(o))
    return bisect(o[:half], pts, size) + bisect(o[half:], pts, size)
def count_steps(atoms, retile):
    xs = (x[atoms] / L % 1.0) * L
    groups = []
    for t0 in range(0, N, 64):
        idx = np.arange(t0, min(t0 + 64, N))
        if retile:
            pts = np.zeros((N, 3)); pts[idx] = unwrap(xs[idx])
            groups += bisect(idx, pts, 8)
        else:
            groups += [idx[k:k + 8] for k in range(0, len(idx), 8)]
    c, h = [], []
    for g in groups:
        q = unwrap(xs[g]); lo, hi = q.min(0), q.max(0); c.append(0.5 * (lo + hi)); h.append(0.5 * (hi - lo))
    c, h = np.array(c), np.array(h)
    steps = 0; inside = 0
    for ic in range(len(groups)):
        d = c[ic] - c[ic:]; d -= L * np.round(d / L)
        g = np.maximum(np.abs(d) - h[ic] - h[ic:], 0.0)
        near = np.nonzero((g * g).sum(-1) < RC * RC)[0]
        steps += len(near)
    return steps, float(np.mean(h.sum(1))) 
for cells, curve in ((16, 'morton'), (8, 'morton'), (8, 'hilbert'), (16, 'hilbert'), (32, 'hilbert')):
    a = order_atoms(cells, curve)
    for retile in (False, True):
        s, hh = count_steps(a, retile)
        print('cells %2d %-7s retile %-5s: steps %6d  mean box half-extent sum %.3f' % (cells, curve, retile, s, hh), flush=True)
