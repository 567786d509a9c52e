"""Known-answer values for the Amber -> System conversion of the host-guest system.

Reads /root/reference/openmmtools/data/cb7-b2/complex-explicit.prmtop with a parser of its own (fixed-width %FLAG sections;
nothing from openmmtools_amd.amber), prints the raw fields behind three 1-4 exceptions (per-dihedral SCEE / SCNB branch of
amber.py) and one GAFF improper, and the values OpenMM's AmberPrmtopFile.createSystem derives from them:
    chargeProd = q_i q_j / 18.2223^2 / SCEE[t]          [e^2]
    sigma      = (sigma_i + sigma_j) / 2,  sigma_t = (A_tt / B_tt)^(1/6) * 0.1 nm      (r_min = (2 A / B)^(1/6), sigma = r_min 2^(-1/6))
    epsilon    = sqrt(eps_i eps_j) / SCNB[t],  eps_t = B_tt^2 / (4 A_tt) * 4.184 kJ/mol
    improper   = (i, j, |k|, |l|, periodicity, phase, k * 4.184)
The literals in tests/test_amber_kat.py are this script's output (build container only; the test itself reads the committed
.npz the product loads)."""
import math
import re

PATH = '/root/reference/openmmtools/data/cb7-b2/complex-explicit.prmtop'
sec, cur, fmt = {}, None, None
for line in open(PATH):
    if line.startswith('%FLAG'):
        cur = line.split()[1]; sec[cur] = []; continue
    if line.startswith('%FORMAT'):
        m = re.match(r'%FORMAT\((\d+)([aEI])(\d+)', line); fmt = (m.group(2), int(m.group(3))); continue
    if line.startswith('%') or cur is None:
        continue
    s = line.rstrip('\n')
    for k in range(0, len(s), fmt[1]):
        t = s[k:k + fmt[1]]
        if t.strip():
            sec[cur].append(t if fmt[0] == 'a' else (int(t) if fmt[0] == 'I' else float(t)))
ntypes = sec['POINTERS'][1]
q, tix, nbi = sec['CHARGE'], sec['ATOM_TYPE_INDEX'], sec['NONBONDED_PARM_INDEX']
A, B = sec['LENNARD_JONES_ACOEF'], sec['LENNARD_JONES_BCOEF']
scee, scnb = sec['SCEE_SCALE_FACTOR'], sec['SCNB_SCALE_FACTOR']


def lj(i):
    t = tix[i]
    k = nbi[ntypes * (t - 1) + (t - 1)] - 1
    if A[k] == 0.0 or B[k] == 0.0:
        return 0.1, 0.0, A[k], B[k]
    return (A[k] / B[k]) ** (1.0 / 6.0) * 0.1, B[k] * B[k] / (4.0 * A[k]) * 4.184, A[k], B[k]


seen, picks = set(), []
want = [lambda i, l: i < 126 and l < 126, lambda i, l: 126 <= i < 156 and 126 <= l < 156 and sec['ATOM_NAME'][l].strip().startswith('H'),
        lambda i, l: 126 <= i < 156 and 126 <= l < 156 and not sec['ATOM_NAME'][i].strip().startswith('H') and not sec['ATOM_NAME'][l].strip().startswith('H')]
for flag in ('DIHEDRALS_INC_HYDROGEN', 'DIHEDRALS_WITHOUT_HYDROGEN'):
    a = sec[flag]
    for k in range(0, len(a), 5):
        i, l3, l4, t = a[k] // 3, a[k + 2], a[k + 3], a[k + 4] - 1
        if l3 < 0 or l4 < 0:
            continue
        l = l4 // 3
        for w in list(want):
            if w(i, l) and (min(i, l), max(i, l)) not in seen and lj(i)[1] > 0 and lj(l)[1] > 0:
                seen.add((min(i, l), max(i, l))); picks.append((i, l, t)); want.remove(w); break
for i, l, t in picks:
    si, ei, Ai, Bi = lj(i); sl, el, Al, Bl = lj(l)
    print('pair (%d, %d) dihedral type %d: CHARGE %r %r  A/B_i %r %r  A/B_l %r %r  SCEE %r SCNB %r' % (i, l, t, q[i], q[l], Ai, Bi, Al, Bl, scee[t], scnb[t]))
    print('    -> chargeProd %.12e  sigma %.12e  epsilon %.12e' % (q[i] * q[l] / 18.2223 ** 2 / scee[t], 0.5 * (si + sl), math.sqrt(ei * el) / scnb[t]))
a = sec['DIHEDRALS_WITHOUT_HYDROGEN']
for k in range(0, len(a), 5):
    if a[k + 3] < 0 and sec['DIHEDRAL_FORCE_CONSTANT'][a[k + 4] - 1] != 0.0:
        t = a[k + 4] - 1
        print('improper (%d, %d, %d, %d) type %d: k %r kcal/mol, n %r, phase %r -> k %.10f kJ/mol' % (
            a[k] // 3, a[k + 1] // 3, abs(a[k + 2]) // 3, abs(a[k + 3]) // 3, t, sec['DIHEDRAL_FORCE_CONSTANT'][t],
            sec['DIHEDRAL_PERIODICITY'][t], sec['DIHEDRAL_PHASE'][t], sec['DIHEDRAL_FORCE_CONSTANT'][t] * 4.184))
        break
