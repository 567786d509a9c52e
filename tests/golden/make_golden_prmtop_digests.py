"""Digests of the three explicit-solvent benchmark systems straight from the reference's Amber files, by a reader of its own (nothing
of openmmtools_amd/amber.py): per system the atom count, sums and extrema of charges / masses / Lennard-Jones sigma and epsilon, counts
and parameter sums of the bonds, angles and dihedrals that survive HBonds constraints + rigid water, the box and a position checksum.
The stored .npz systems of openmmtools_amd/data (made by tools/convert_amber.py) are held to these in tests/test_testsystem_defaults.py.

The alanine dipeptide system is, in addition, pinned field by field to the System OpenMM built from the same prmtop
(tests/test_openmm_fixture.py) -- that is what calibrates the unit conversions used here; CB7:B2 and DHFR have only this and
tests/test_amber_kat.py.            usage: python tests/golden/make_golden_prmtop_digests.py"""
import json
import os
import re

import numpy as np

REF = '/root/reference/openmmtools/data'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'prmtop_digests.json')
JOBS = [('alanine-dipeptide-explicit', 'alanine-dipeptide-explicit/alanine-dipeptide.prmtop', 'alanine-dipeptide-explicit/alanine-dipeptide.crd'),
        ('cb7-b2-explicit', 'cb7-b2/complex-explicit.prmtop', 'cb7-b2/complex-explicit.inpcrd'),
        ('dhfr-explicit', 'dhfr/JAC.prmtop', 'dhfr/JAC.inpcrd')]
KCAL, ANG = 4.184, 0.1
AMBER_CHARGE = 18.2223                      # sqrt(332.0522...) : prmtop charges are in units of e x 18.2223


def sections(path):
    out, name, fmt = {}, None, None
    for line in open(path):
        if line.startswith('%FLAG'):
            name = line.split()[1]; out[name] = []
        elif line.startswith('%FORMAT'):
            m = re.match(r'%FORMAT\((\d+)([aIEe])(\d+)', line)
            fmt = (m.group(2), int(m.group(3)))
        elif name and not line.startswith('%'):
            kind, w = fmt
            text = line.rstrip('\n')
            for k in range(0, len(text), w):
                tok = text[k:k + w]
                if tok.strip() == '' and kind != 'a':
                    continue
                out[name].append(tok if kind == 'a' else (int(tok) if kind == 'I' else float(tok)))
    return out


def digest(top, crd):
    s = sections(top)
    n = s['POINTERS'][0]
    ntypes = s['POINTERS'][1]
    q = np.array(s['CHARGE']) / AMBER_CHARGE
    m = np.array(s['MASS'])
    # per-atom Lennard-Jones from the diagonal of the A / B tables: sigma = (A/B)^(1/6), epsilon = B^2 / (4 A)
    t = np.array(s['ATOM_TYPE_INDEX']) - 1
    idx = np.array(s['NONBONDED_PARM_INDEX'])[t * ntypes + t] - 1
    A, B = np.array(s['LENNARD_JONES_ACOEF'])[idx], np.array(s['LENNARD_JONES_BCOEF'])[idx]
    with np.errstate(divide='ignore', invalid='ignore'):
        sigma = np.where(B > 0, (A / np.where(B > 0, B, 1.0)) ** (1.0 / 6.0), 0.0) * ANG
        eps = np.where(A > 0, B * B / (4.0 * np.where(A > 0, A, 1.0)), 0.0) * KCAL
    names = [a.strip() for a in s['ATOM_NAME']]
    # bonds: triples (3 i, 3 j, type); with HBonds every bond to a hydrogen is a constraint, and rigid water constrains the rest of a water
    res_ptr = np.array(s['RESIDUE_POINTER']) - 1
    res_of = np.searchsorted(res_ptr, np.arange(n), side='right') - 1
    water = np.array([lab.strip() in ('WAT', 'HOH') for lab in s['RESIDUE_LABEL']])[res_of]
    bk, br = np.array(s['BOND_FORCE_CONSTANT']), np.array(s['BOND_EQUIL_VALUE'])
    bh = np.array(s['BONDS_INC_HYDROGEN']).reshape(-1, 3)
    ba = np.array(s['BONDS_WITHOUT_HYDROGEN']).reshape(-1, 3)
    free_bonds = [(b[0] // 3, b[1] // 3, b[2] - 1) for b in ba if not (water[b[0] // 3] and water[b[1] // 3])]
    n_constraints = len(bh) + sum(1 for b in ba if water[b[0] // 3] and water[b[1] // 3])
    ak, at = np.array(s['ANGLE_FORCE_CONSTANT']), np.array(s['ANGLE_EQUIL_VALUE'])
    ang = np.concatenate([np.array(s['ANGLES_INC_HYDROGEN']).reshape(-1, 4), np.array(s['ANGLES_WITHOUT_HYDROGEN']).reshape(-1, 4)])
    free_angles = [a for a in ang if not water[a[1] // 3]]
    dk, dn, dp = np.array(s['DIHEDRAL_FORCE_CONSTANT']), np.array(s['DIHEDRAL_PERIODICITY']), np.array(s['DIHEDRAL_PHASE'])
    dih = np.concatenate([np.array(s['DIHEDRALS_INC_HYDROGEN']).reshape(-1, 5), np.array(s['DIHEDRALS_WITHOUT_HYDROGEN']).reshape(-1, 5)])
    # exceptions: every entry of the excluded-atoms list is one; the 1-4 pairs among them (end atoms of the dihedrals whose third and
    # fourth pointers are not negative: negative third = "1-4 already counted or inside a small ring", negative fourth = improper) keep
    # q_i q_l / SCEE and the type pair's Lennard-Jones epsilon / SCNB (per dihedral type when the prmtop has the sections, else 1.2 / 2.0)
    n_exceptions = int(sum(1 for e in s['EXCLUDED_ATOMS_LIST'] if e > 0))
    scee = np.array(s['SCEE_SCALE_FACTOR']) if 'SCEE_SCALE_FACTOR' in s else np.full(len(dk), 1.2)
    scnb = np.array(s['SCNB_SCALE_FACTOR']) if 'SCNB_SCALE_FACTOR' in s else np.full(len(dk), 2.0)
    Aall, Ball, nbidx = np.array(s['LENNARD_JONES_ACOEF']), np.array(s['LENNARD_JONES_BCOEF']), np.array(s['NONBONDED_PARM_INDEX'])
    seen, q14, e14, s14 = set(), 0.0, 0.0, 0.0
    for d in dih:
        if d[2] < 0 or d[3] < 0:
            continue
        i, l, ty = d[0] // 3, d[3] // 3, d[4] - 1
        if (min(i, l), max(i, l)) in seen:
            continue
        seen.add((min(i, l), max(i, l)))
        q14 += q[i] * q[l] / scee[ty]
        k = nbidx[t[i] * ntypes + t[l]] - 1
        if Aall[k] > 0 and Ball[k] > 0:
            e14 += Ball[k] ** 2 / (4.0 * Aall[k]) * KCAL / scnb[ty]
            s14 += (Aall[k] / Ball[k]) ** (1.0 / 6.0) * ANG
    if open(crd, 'rb').read(3) == b'CDF':                       # Amber NetCDF restart (JAC.inpcrd): scipy's NetCDF-3 reader
        from scipy.io import netcdf_file
        with netcdf_file(crd, 'r', mmap=False) as nc:
            pos = np.array(nc.variables['coordinates'][:], dtype=np.float64).reshape(n, 3) * ANG
            box = np.array(nc.variables['cell_lengths'][:], dtype=np.float64) * ANG
            has_vel = 'velocities' in nc.variables
    else:
        lines = open(crd).read().split('\n')
        vals = []
        for line in lines[2:]:
            vals += [float(line[k:k + 12]) for k in range(0, len(line.rstrip()), 12) if line[k:k + 12].strip()]
        vals = np.array(vals)
        has_vel = len(vals) >= 6 * n + 3
        pos = vals[:3 * n].reshape(n, 3) * ANG
        box = (vals[6 * n:6 * n + 3] if has_vel else vals[3 * n:3 * n + 3]) * ANG
    return dict(n_atoms=int(n), charge_sum=float(q.sum()), charge_abs_sum=float(np.abs(q).sum()), mass_sum=float(m.sum()),
                sigma_sum_where_epsilon_nonzero=float(sigma[eps > 0].sum()), n_epsilon_nonzero=int((eps > 0).sum()), epsilon_sum=float(eps.sum()), sigma_max=float(sigma.max()), epsilon_max=float(eps.max()),
                n_constraints=int(n_constraints), n_bonds=len(free_bonds),
                bond_k_sum=float(sum(2.0 * bk[t] * KCAL / ANG ** 2 for _, _, t in free_bonds)), bond_r0_sum=float(sum(br[t] * ANG for _, _, t in free_bonds)),
                n_angles=len(free_angles), angle_k_sum=float(sum(2.0 * ak[a[3] - 1] * KCAL for a in free_angles)),
                angle_theta_sum=float(sum(at[a[3] - 1] for a in free_angles)),
                n_dihedrals=int(len(dih)), dihedral_k_sum=float(sum(dk[d[4] - 1] * KCAL for d in dih)),
                n_dihedrals_nonzero=int(sum(1 for d in dih if dk[d[4] - 1] != 0.0)),
                dihedral_periodicity_sum_nonzero=float(sum(abs(dn[d[4] - 1]) for d in dih if dk[d[4] - 1] != 0.0)),
                dihedral_phase_sum_nonzero=float(sum(dp[d[4] - 1] for d in dih if dk[d[4] - 1] != 0.0)),
                n_exceptions=n_exceptions, n_14=len(seen), charge_product_14_sum=float(q14), epsilon_14_sum=float(e14), sigma_14_sum_where_epsilon_nonzero=float(s14),
                box=[float(b) for b in box], position_sum=float(pos.sum()), position_abs_sum=float(np.abs(pos).sum()), has_velocities=bool(has_vel),
                first_atoms=names[:6])


if __name__ == '__main__':
    out = {name: digest(os.path.join(REF, top), os.path.join(REF, crd)) for name, top, crd in JOBS}
    with open(OUT, 'w') as fh:
        json.dump(out, fh, indent=1)
    for k, v in out.items():
        print(k, {a: (round(b, 6) if isinstance(b, float) else b) for a, b in v.items() if a not in ('first_atoms',)})
