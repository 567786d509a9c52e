"""Amber prmtop / inpcrd readers producing the System shim.

Stand-in for ``openmm.app.AmberPrmtopFile(...).createSystem(constraints=HBonds, rigidWater=True,
nonbondedMethod=PME)`` as called by openmmtools/testsystems.py:3504-3505, 3826-3835, 3900-3901
(OpenMM itself is not importable here).  Conversion rules: charges / 18.2223,
Angstrom -> nm, kcal -> kJ; Amber E = K (r - r0)^2 => harmonic k = 2K; torsion k = PK, periodicity
|PN|; per-type LJ from the diagonal A/B coefficients (eps = B^2/4A, rmin = (2A/B)^(1/6),
sigma = rmin 2^(-1/6)); 1-4 exceptions qq/SCEE, sqrt(eps eps)/SCNB with defaults 1.2 / 2.0;
EXCLUDED_ATOMS_LIST => zeroed exceptions; every bond involving hydrogen becomes a distance
constraint at its equilibrium length and is dropped from the bond force.
"""
import re
import numpy as np
from .system import (System, NonbondedForce, HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce,
                     CMMotionRemover)

KCAL = 4.184
AMBER_CHARGE = 18.2223


def read_prmtop(path):
    sections, fmt, name = {}, None, None
    with open(path) as fh:
        for line in fh:
            if line.startswith('%VERSION'):
                continue
            if line.startswith('%FLAG'):
                name = line.split()[1]
                sections[name] = []
                fmt = None
            elif line.startswith('%FORMAT'):
                m = re.match(r'%FORMAT\((\d+)([aIE])(\d+)(?:\.(\d+))?\)', line.strip())
                fmt = (int(m.group(1)), m.group(2), int(m.group(3)))
            elif name is not None and fmt is not None:
                line = line.rstrip('\n')
                count, kind, width = fmt
                for k in range(0, len(line), width):
                    tok = line[k:k + width]
                    if not tok.strip() and kind != 'a':
                        continue
                    if kind == 'a':
                        sections[name].append(tok.strip())
                    elif kind == 'I':
                        sections[name].append(int(tok))
                    else:
                        sections[name].append(float(tok))
    return sections


def read_inpcrd(path, natom):
    """ASCII inpcrd: returns positions [nm], velocities or None, box edge lengths [nm] or None."""
    with open(path, 'rb') as fh:
        head = fh.read(4)
    if head[:3] == b'CDF':
        return read_netcdf_restart(path)
    with open(path) as fh:
        lines = fh.read().split('\n')
    n = int(lines[1].split()[0])
    assert n == natom
    vals = []
    for line in lines[2:]:
        for k in range(0, len(line.rstrip()), 12):
            vals.append(float(line[k:k + 12]))
    pos = np.array(vals[:3 * n]).reshape(n, 3) * 0.1
    rest = vals[3 * n:]
    vel, box = None, None
    if len(rest) >= 3 * n + 3:
        vel = np.array(rest[:3 * n]).reshape(n, 3) * 20.455 * 0.1
        rest = rest[3 * n:]
    if len(rest) >= 3:
        box = np.array(rest[:3]) * 0.1
    return pos, vel, box


def read_netcdf_restart(path):
    from scipy.io import netcdf_file
    f = netcdf_file(path, 'r', mmap=False)
    pos = np.array(f.variables['coordinates'][:], dtype=np.float64) * 0.1
    vel = None
    if 'velocities' in f.variables:
        v = f.variables['velocities']
        scale = getattr(v, 'scale_factor', 20.455)
        vel = np.array(v[:], dtype=np.float64) * float(scale) * 0.1
    box = np.array(f.variables['cell_lengths'][:], dtype=np.float64) * 0.1 if 'cell_lengths' in f.variables else None
    f.close()
    return pos, vel, box


def create_system(prm, removeCMMotion=True):
    """prmtop sections -> System with constraints=HBonds, rigidWater=True semantics."""
    ptr = prm['POINTERS']
    natom, ntypes = ptr[0], ptr[1]
    system = System()
    for m in prm['MASS'][:natom]:
        system.addParticle(m)
    # --- bonds ---
    bf = HarmonicBondForce()
    bk, br = prm['BOND_FORCE_CONSTANT'], prm['BOND_EQUIL_VALUE']

    def bond_list(flag):
        a = prm[flag]
        return [(a[k] // 3, a[k + 1] // 3, a[k + 2] - 1) for k in range(0, len(a), 3)]
    # constraints=HBonds: the prmtop itself lists the bonds that involve hydrogen (robust against
    # hydrogen-mass-repartitioned topologies such as JAC.prmtop, where H masses are 3.024)
    for (i, j, t) in bond_list('BONDS_INC_HYDROGEN'):
        system.addConstraint(i, j, br[t] * 0.1)
    for (i, j, t) in bond_list('BONDS_WITHOUT_HYDROGEN'):
        bf.addBond(i, j, br[t] * 0.1, 2.0 * bk[t] * KCAL * 100.0)
    # --- angles ---
    af = HarmonicAngleForce()
    ak, at = prm['ANGLE_FORCE_CONSTANT'], prm['ANGLE_EQUIL_VALUE']
    constrained = set()
    for (i, j, d) in system.constraints:
        constrained.add((min(i, j), max(i, j)))
    for flag in ('ANGLES_INC_HYDROGEN', 'ANGLES_WITHOUT_HYDROGEN'):
        a = prm[flag]
        for k in range(0, len(a), 4):
            i, j, l, t = a[k] // 3, a[k + 1] // 3, a[k + 2] // 3, a[k + 3] - 1
            # rigid water: an H-O-H angle whose three sides are all constrained carries no energy term
            if all((min(p, q), max(p, q)) in constrained for p, q in ((i, j), (j, l), (i, l))):
                continue
            af.addAngle(i, j, l, at[t], 2.0 * ak[t] * KCAL)
    # --- torsions and 1-4 pairs ---
    tf = PeriodicTorsionForce()
    dk, dn, dp = prm['DIHEDRAL_FORCE_CONSTANT'], prm['DIHEDRAL_PERIODICITY'], prm['DIHEDRAL_PHASE']
    scee = prm.get('SCEE_SCALE_FACTOR', None)
    scnb = prm.get('SCNB_SCALE_FACTOR', None)
    one_four = {}
    for flag in ('DIHEDRALS_INC_HYDROGEN', 'DIHEDRALS_WITHOUT_HYDROGEN'):
        a = prm[flag]
        for k in range(0, len(a), 5):
            i, j, l3, l4, t = a[k] // 3, a[k + 1] // 3, a[k + 2], a[k + 3], a[k + 4] - 1
            if dk[t] != 0.0:
                tf.addTorsion(i, j, abs(l3) // 3, abs(l4) // 3, int(abs(round(dn[t]))), dp[t], dk[t] * KCAL)
            if l3 >= 0 and l4 >= 0:
                p, q = i, l4 // 3
                key = (min(p, q), max(p, q))
                if key not in one_four:
                    one_four[key] = (scee[t] if scee else 1.2, scnb[t] if scnb else 2.0)
    # --- nonbonded ---
    nb = NonbondedForce()
    tidx = prm['ATOM_TYPE_INDEX'][:natom]
    nbidx = prm['NONBONDED_PARM_INDEX']
    A, B = prm['LENNARD_JONES_ACOEF'], prm['LENNARD_JONES_BCOEF']
    type_sig, type_eps = {}, {}
    for t in set(tidx):
        k = nbidx[ntypes * (t - 1) + (t - 1)] - 1
        a, b = A[k], B[k]
        if a == 0.0 or b == 0.0:
            type_sig[t], type_eps[t] = 0.1, 0.0         # 1 Angstrom placeholder, no interaction
        else:
            rmin = (2.0 * a / b) ** (1.0 / 6.0)
            type_sig[t] = rmin * 2.0 ** (-1.0 / 6.0) * 0.1
            type_eps[t] = b * b / (4.0 * a) * KCAL
    charges = np.array(prm['CHARGE'][:natom]) / AMBER_CHARGE
    for i in range(natom):
        nb.addParticle(charges[i], type_sig[tidx[i]], type_eps[tidx[i]])
    # exclusions, then 1-4 exceptions
    nex = prm['NUMBER_EXCLUDED_ATOMS'][:natom]
    exl = prm['EXCLUDED_ATOMS_LIST']
    pos, excluded = 0, set()
    for i in range(natom):
        for k in range(nex[i]):
            j = exl[pos + k] - 1
            if j >= 0 and j != i:
                excluded.add((min(i, j), max(i, j)))
        pos += nex[i]
    for key, (se, sn) in one_four.items():
        excluded.add(key)
    for (i, j) in sorted(excluded):
        if (i, j) in one_four:
            se, sn = one_four[(i, j)]
            qi, si, ei = nb.particles[i]
            qj, sj, ej = nb.particles[j]
            nb.addException(i, j, qi * qj / se, 0.5 * (si + sj), np.sqrt(ei * ej) / sn)
        else:
            nb.addException(i, j, 0.0, 0.1, 0.0)
    for f in (bf, af, tf, nb):
        system.addForce(f)
    if removeCMMotion:
        system.addForce(CMMotionRemover(1))
    return system, nb
