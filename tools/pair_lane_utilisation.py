"""Round 6: how many of the pair kernel's 8 x 8 cluster-pair steps would a 4-atom j half-cluster save?  CPU model of the
device's list geometry on the headline system: molecules ordered along a Morton curve (their atoms stay together), 8-atom clusters of that
order, 64-atom tiles, an entry (tile, j cluster) per j cluster whose bounding box is within the Coulomb range of the tile's, one
cluster-pair step per i cluster of the tile whose box is within range of the j cluster's (jc >= ic: every pair once).  Counted:
steps, lane pairs inside the range, and -- per shell of box-to-box distance -- the steps in which only ONE 4-atom half of the j cluster is
in range of the i cluster's box (a step that a half-cluster entry would turn into half a step IF it found a partner to share lanes with).
usage: python tools/pair_lane_utilisation.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmmtools_amd import testsystems as ts

RC = 1.126
al = ts.AlanineDipeptideExplicit()
x = np.asarray(al.positions, dtype=np.float64)
L = np.diag(al.system.getDefaultPeriodicBoxVectors()).astype(np.float64)
N = x.shape[0]
# molecules: solute (22 atoms) + 749 waters of 3
mol_first = [0] + list(range(22, N, 3))
mol_size = [22] + [3] * ((N - 22) // 3)
def morton(c):
    k = 0
    for b in range(10):
        for a in range(3):
            k |= ((int(c[a]) >> b) & 1) << (3 * b + a)
    return k
cells = 16
keys = []
for f in mol_first:
    frac = (x[f] / L) % 1.0
    keys.append(morton((frac * cells).astype(int)))
order = np.argsort(np.array(keys), kind='stable')
atoms = np.concatenate([np.arange(mol_first[m], mol_first[m] + mol_size[m]) for m in order])
xs = (x[atoms] / L % 1.0) * L
npad = (N + 63) // 64 * 64
ncl = npad // 8
def boxes(idx_groups):
    c, h = [], []
    for g in idx_groups:
        p = xs[g]
        # minimum-image centre: unwrap around the first atom
        d = p - p[0]; d -= L * np.round(d / L); q = p[0] + d
        lo, hi = q.min(0), q.max(0)
        c.append(0.5 * (lo + hi)); h.append(0.5 * (hi - lo))
    return np.array(c), np.array(h)
cl_idx = [np.arange(8 * c, min(8 * c + 8, N)) for c in range(ncl) if 8 * c < N]
ncl = len(cl_idx)
cc, ch = boxes(cl_idx)
half_idx = [[g[:4], g[4:]] for g in cl_idx]
hc = [boxes([h for h in hh if len(h)]) for hh in half_idx]
def box_dist2(c1, h1, c2, h2):
    d = c1 - c2; d -= L * np.round(d / L)
    g = np.maximum(np.abs(d) - h1 - h2, 0.0)
    return (g * g).sum(-1)
steps = lanes_in = half_steps = 0
shell = {}
for ic in range(ncl):
    d2 = box_dist2(cc[ic], ch[ic], cc[ic:], ch[ic:])
    for off in np.nonzero(d2 < RC * RC)[0]:
        jc = ic + off
        steps += 1
        pi, pj = xs[cl_idx[ic]], xs[cl_idx[jc]]
        d = pi[:, None, :] - pj[None, :, :]; d -= L * np.round(d / L)
        r2 = (d * d).sum(-1)
        inside = r2 < RC * RC
        if jc == ic: inside = np.triu(inside, 1)
        lanes_in += int(inside.sum())
        hcj, hhj = hc[jc]
        n_half = sum(1 for q in range(len(hcj)) if box_dist2(cc[ic], ch[ic], hcj[q], hhj[q]) < RC * RC)
        key = min(int(np.sqrt(d2[off]) / 0.2), 5)
        s = shell.setdefault(key, [0, 0, 0])
        s[0] += 1; s[1] += int(inside.sum()); s[2] += (n_half == 1)
        half_steps += (n_half == 1)
print('clusters %d  cluster-pair steps %d  lane pairs inside %.3f of 64 per step (%.1f %% useful lanes)' % (ncl, steps, lanes_in / steps, 100 * lanes_in / steps / 64))
print('steps in which only one 4-atom j half is in range of the i cluster box: %d = %.1f %% of the steps' % (half_steps, 100 * half_steps / steps))
print('upper bound of the saving if every such step found a partner: %.1f %% of the steps' % (50 * half_steps / steps))
for k in sorted(shell):
    s = shell[k]
    print('  box gap %.1f-%.1f nm: %6d steps (%.1f %%), useful lanes %.1f %%, one-half-only steps %.1f %%' % (0.2 * k, 0.2 * k + 0.2, s[0], 100 * s[0] / steps, 100 * s[1] / s[0] / 64, 100 * s[2] / s[0]))

# ---- tighter list tests (all necessary conditions for a pair inside the range, i.e. exact lists): per step, does it survive
#   A: some j ATOM within range of the i cluster's BOX;  B: some i ATOM within range of the j cluster's BOX;  S: bounding spheres
def point_box_d2(p, c, h):
    d = p - c; d -= L * np.round(d / L)
    g = np.maximum(np.abs(d) - h, 0.0)
    return (g * g).sum(-1)
cent = np.array([xs[g].mean(0) for g in cl_idx])     # (clusters do not straddle the box after the molecule-wise wrap in this model: unwrap as above)
def unwrapped(g):
    p = xs[g]; d = p - p[0]; d -= L * np.round(d / L); return p[0] + d
cent = np.array([unwrapped(g).mean(0) for g in cl_idx]); rad = np.array([np.linalg.norm(unwrapped(g) - unwrapped(g).mean(0), axis=1).max() for g in cl_idx])
surv = dict(A=0, B=0, AB=0, S=0, ABS=0, exact=0)
for ic in range(ncl):
    d2 = box_dist2(cc[ic], ch[ic], cc[ic:], ch[ic:])
    for off in np.nonzero(d2 < RC * RC)[0]:
        jc = ic + off
        pi, pj = xs[cl_idx[ic]], xs[cl_idx[jc]]
        a = bool((point_box_d2(pj, cc[ic], ch[ic]) < RC * RC).any())
        b = bool((point_box_d2(pi, cc[jc], ch[jc]) < RC * RC).any())
        dcen = cent[ic] - cent[jc]; dcen -= L * np.round(dcen / L)
        s_ = np.linalg.norm(dcen) - rad[ic] - rad[jc] < RC
        d = pi[:, None, :] - pj[None, :, :]; d -= L * np.round(d / L)
        ex = bool(((d * d).sum(-1) < RC * RC).any())
        surv['A'] += a; surv['B'] += b; surv['AB'] += a and b; surv['S'] += s_; surv['ABS'] += a and b and s_; surv['exact'] += ex
print('steps surviving a tighter (still exact) list test, of %d box-box steps:' % steps)
for k in ('A', 'B', 'AB', 'S', 'ABS', 'exact'):
    print('  %-5s %6d  (%.1f %%)' % (k, surv[k], 100.0 * surv[k] / steps))
