"""Minimal stand-in for ``openmm.System`` and the Force classes the benchmark test systems use.

The reference's input contract is an ``openmm.System`` wrapped in a ``ThermodynamicState``
(openmmtools/states.py:1494 hashes its XML).  OpenMM is not importable in the build
environment, so this module mirrors the subset of the OpenMM API (same method names and
argument order, md-unit floats instead of Quantities) that openmmtools/testsystems.py uses
for HarmonicOscillator (:761-802), LennardJonesFluid (:1939-2030) and the Amber-built
explicit-solvent systems (:3496-3527, :3818-3857, :3892-3923).  ``system_to_desc`` flattens a
System into the arrays behind ``remd_system_desc`` (include/remd_hip.h).  A real
``openmm.System`` can be converted with ``from_openmm`` where OpenMM exists.
"""
import hashlib
import math
import os
import numpy as np


class Force:
    def __init__(self):
        self._group = 0

    def getForceGroup(self):
        return self._group

    def setForceGroup(self, g):
        self._group = int(g)


class HarmonicBondForce(Force):
    def __init__(self):
        super().__init__()
        self.bonds = []

    def addBond(self, i, j, length, k):
        self.bonds.append((int(i), int(j), float(length), float(k)))
        return len(self.bonds) - 1

    def getNumBonds(self):
        return len(self.bonds)

    def getBondParameters(self, idx):
        return self.bonds[idx]


class HarmonicAngleForce(Force):
    def __init__(self):
        super().__init__()
        self.angles = []

    def addAngle(self, i, j, k, angle, kf):
        self.angles.append((int(i), int(j), int(k), float(angle), float(kf)))
        return len(self.angles) - 1

    def getNumAngles(self):
        return len(self.angles)

    def getAngleParameters(self, idx):
        return self.angles[idx]


class PeriodicTorsionForce(Force):
    def __init__(self):
        super().__init__()
        self.torsions = []

    def addTorsion(self, i, j, k, l, periodicity, phase, kf):
        self.torsions.append((int(i), int(j), int(k), int(l), int(periodicity), float(phase), float(kf)))
        return len(self.torsions) - 1

    def getNumTorsions(self):
        return len(self.torsions)

    def getTorsionParameters(self, idx):
        return self.torsions[idx]


class NonbondedForce(Force):
    NoCutoff, CutoffNonPeriodic, CutoffPeriodic, Ewald, PME = 0, 1, 2, 3, 4

    def __init__(self):
        super().__init__()
        self.particles = []
        self.exceptions = []
        self._method = NonbondedForce.NoCutoff
        self._cutoff = 1.0
        self._use_switch = False
        self._switch = -1.0
        self._dispersion = True
        self._rf_dielectric = 78.3
        self._ewald_tol = 5e-4
        self._pme_params = None
        self._recip_group = -1

    def getReciprocalSpaceForceGroup(self):
        return self._recip_group

    def setReciprocalSpaceForceGroup(self, g):
        """-1 (default): the reciprocal-space part acts in the force group of the force itself."""
        self._recip_group = int(g)

    def addParticle(self, charge, sigma, epsilon):
        self.particles.append((float(charge), float(sigma), float(epsilon)))
        return len(self.particles) - 1

    def getNumParticles(self):
        return len(self.particles)

    def getParticleParameters(self, idx):
        return self.particles[idx]

    def setParticleParameters(self, idx, charge, sigma, epsilon):
        self.particles[idx] = (float(charge), float(sigma), float(epsilon))

    def addException(self, i, j, chargeProd, sigma, epsilon, replace=False):
        i, j = int(i), int(j)
        for n, e in enumerate(self.exceptions):
            if {e[0], e[1]} == {i, j}:
                if not replace:
                    raise ValueError('exception between %d and %d already exists' % (i, j))
                self.exceptions[n] = (i, j, float(chargeProd), float(sigma), float(epsilon))
                return n
        self.exceptions.append((i, j, float(chargeProd), float(sigma), float(epsilon)))
        return len(self.exceptions) - 1

    def getNumExceptions(self):
        return len(self.exceptions)

    def getExceptionParameters(self, idx):
        return self.exceptions[idx]

    def setNonbondedMethod(self, m):
        self._method = int(m)

    def getNonbondedMethod(self):
        return self._method

    def setCutoffDistance(self, d):
        self._cutoff = float(d)

    def getCutoffDistance(self):
        return self._cutoff

    def setUseSwitchingFunction(self, flag):
        self._use_switch = bool(flag)

    def getUseSwitchingFunction(self):
        return self._use_switch

    def setSwitchingDistance(self, d):
        self._switch = float(d)

    def getSwitchingDistance(self):
        return self._switch

    def setUseDispersionCorrection(self, flag):
        self._dispersion = bool(flag)

    def getUseDispersionCorrection(self):
        return self._dispersion

    def setReactionFieldDielectric(self, e):
        self._rf_dielectric = float(e)

    def getReactionFieldDielectric(self):
        return self._rf_dielectric

    def setEwaldErrorTolerance(self, tol):
        self._ewald_tol = float(tol)

    def getEwaldErrorTolerance(self):
        return self._ewald_tol

    def setPMEParameters(self, alpha, nx, ny, nz):
        self._pme_params = (float(alpha), int(nx), int(ny), int(nz))

    def usesPeriodicBoundaryConditions(self):
        return self._method in (NonbondedForce.CutoffPeriodic, NonbondedForce.Ewald, NonbondedForce.PME)


class CustomExternalForce(Force):
    """Only the harmonic-well expression of testsystems.HarmonicOscillator (testsystems.py:779-786)."""

    HARMONIC_EXPRESSION = ('(K/2.0) * ((x-x0)^2 + y^2 + z^2) + U0;'
                           'K = testsystems_HarmonicOscillator_K;'
                           'x0 = testsystems_HarmonicOscillator_x0;'
                           'U0 = testsystems_HarmonicOscillator_U0;')

    def __init__(self, energy_expression):
        super().__init__()
        if energy_expression.replace(' ', '') != self.HARMONIC_EXPRESSION.replace(' ', ''):
            raise NotImplementedError('only the testsystems.HarmonicOscillator expression is supported')
        self.energy_expression = energy_expression
        self.globals = {}
        self.particles = []

    def addGlobalParameter(self, name, value):
        self.globals[name] = float(value)

    def getGlobalParameter(self, name):
        return self.globals[name]

    def addParticle(self, index, params=()):
        self.particles.append(int(index))

    def getNumParticles(self):
        return len(self.particles)


class GBSAOBCForce(Force):
    """openmm.GBSAOBCForce: OBC2 Born radii + ACE surface term (the form the reference's alchemical factory spells out, alchemy.py:2144-2225).
    NoCutoff only."""
    NoCutoff, CutoffNonPeriodic, CutoffPeriodic = 0, 1, 2

    def __init__(self):
        super().__init__()
        self.particles = []                 # (charge, radius nm, scale)
        self._solvent, self._solute, self._sa, self._method = 78.5, 1.0, 2.25936, 0

    def addParticle(self, charge, radius, scalingFactor):
        self.particles.append((float(charge), float(radius), float(scalingFactor)))
        return len(self.particles) - 1

    def getNumParticles(self): return len(self.particles)
    def getParticleParameters(self, idx): return self.particles[idx]
    def setParticleParameters(self, idx, charge, radius, scalingFactor): self.particles[idx] = (float(charge), float(radius), float(scalingFactor))
    def getSolventDielectric(self): return self._solvent
    def setSolventDielectric(self, v): self._solvent = float(v)
    def getSoluteDielectric(self): return self._solute
    def setSoluteDielectric(self, v): self._solute = float(v)
    def getSurfaceAreaEnergy(self): return self._sa
    def setSurfaceAreaEnergy(self, v): self._sa = float(v)
    def getNonbondedMethod(self): return self._method
    def setNonbondedMethod(self, m): self._method = int(m)


class CMMotionRemover(Force):
    def __init__(self, frequency=1):
        super().__init__()
        self.frequency = int(frequency)

    def getFrequency(self):
        return self.frequency


class System:
    def __init__(self):
        self.masses = []
        self.forces = []
        self.constraints = []
        self._box = ((2.0, 0.0, 0.0), (0.0, 2.0, 0.0), (0.0, 0.0, 2.0))
        self.alchemical_region = None     # set by alchemy.AbsoluteAlchemicalFactory (one region on the pair kernels' own path)
        self.alchemical_regions = None    # ... general regions: the factory's force split (alchemical_region_terms, csrc/alch_regions.hip)

    def addParticle(self, mass):
        self.masses.append(float(mass))
        return len(self.masses) - 1

    def getNumParticles(self):
        return len(self.masses)

    def getParticleMass(self, i):
        return self.masses[i]

    def addForce(self, force):
        self.forces.append(force)
        return len(self.forces) - 1

    def getNumForces(self):
        return len(self.forces)

    def getForce(self, i):
        return self.forces[i]

    def getForces(self):
        return list(self.forces)

    def addConstraint(self, i, j, distance):
        self.constraints.append((int(i), int(j), float(distance)))
        return len(self.constraints) - 1

    def getNumConstraints(self):
        return len(self.constraints)

    def getConstraintParameters(self, idx):
        return self.constraints[idx]

    def setDefaultPeriodicBoxVectors(self, a, b, c):
        self._box = (tuple(float(v) for v in a), tuple(float(v) for v in b), tuple(float(v) for v in c))

    def getDefaultPeriodicBoxVectors(self):
        return np.array(self._box, dtype=np.float64)

    def usesPeriodicBoundaryConditions(self):
        return any(getattr(f, 'usesPeriodicBoundaryConditions', lambda: False)() for f in self.forces)

    def fingerprint(self):
        """Stable hash of the system's content (stand-in for the XML hash, states.py:1492-1495)."""
        hsh = hashlib.sha1()

        def feed(d):
            for key in sorted(d):
                v = d[key]
                hsh.update(key.encode())
                if isinstance(v, dict):
                    feed(v)
                else:
                    hsh.update(np.ascontiguousarray(v).tobytes() if isinstance(v, np.ndarray) else repr(v).encode())
        feed(system_to_desc(self))
        return hsh.hexdigest()


# ---------------------------------------------------------------------------------------------

def ewald_parameters(cutoff, tolerance, box):
    """alpha and mesh size from the Ewald error tolerance.

    alpha = sqrt(-ln(2 tol))/r_c (the formula the reference quotes at alchemy.py:1528-1532);
    mesh >= 2 alpha L / (3 tol^(1/5)) per axis, rounded up to a product of 2, 3, 5 (the radices of the
    in-tree FFT).
    """
    alpha = math.sqrt(-math.log(2.0 * tolerance)) / cutoff
    grid = []
    for L in box:
        n = int(math.ceil(2.0 * alpha * L / (3.0 * tolerance ** 0.2)))
        n = max(n, 6)
        while True:
            m = n
            for p in (2, 3, 5):
                while m % p == 0:
                    m //= p
            if m == 1:
                break
            n += 1
        grid.append(n)
    return alpha, grid


def rebalanced_coulomb_cutoff(cutoff, tolerance, box, max_extension=1.25, min_edge=None):
    """Range of the Ewald direct-space sum that the device engine prefers for this box (``ewald_split='auto'``), or ``cutoff``.

    The Ewald sum does not depend on where it is split; OpenMM ties the split to the NonbondedForce cutoff
    (alpha = sqrt(-ln 2 tol) / r_c, the rule quoted at alchemy.py:1528-1532), which for AlanineDipeptideExplicit means a
    75 x 75 x 72 mesh.  On the MI355X the mesh chain (spread -> planes -> gather) is the critical path of an MD step and the
    plane pass quantises badly: an nx x ny plane of complex f32 must fit the LDS several times per CU, and power-of-two lines
    need a third fewer butterfly stages.  So the Coulomb range is stretched, by at most ``max_extension``, to the shortest
    range >= cutoff for which the SAME tolerance rule asks for a mesh whose largest edge is one of the plane-friendly sizes
    below; Lennard-Jones terms keep ``cutoff``.  Measured on 24 x alanine dipeptide: profiles/r04_ewald_split_sweep.txt.
    """
    friendly = (32, 40, 48, 64, 80, 96, 128)
    root = math.sqrt(-math.log(2.0 * tolerance))
    lmax = max(float(L) for L in box)
    # (min_edge: the shortest box edge the run may see -- the smallest starting box of an ensemble, less a margin under a barostat;
    # twice the stretched range must stay below it)
    lmin = min(float(L) for L in box) if min_edge is None else min(float(min_edge), min(float(L) for L in box))
    n_ref = max(ewald_parameters(cutoff, tolerance, box)[1])
    best = cutoff
    for n in friendly:
        if n >= n_ref:
            break
        # largest alpha whose rule mesh is <= n on the longest edge, a hair inside the ceil()
        alpha = (n - 1e-6) * 3.0 * tolerance ** 0.2 / (2.0 * lmax)
        rcc = root / alpha
        if cutoff < rcc <= max_extension * cutoff and 2.0 * rcc < lmin:
            best = rcc
    return best


def _classify_constraints(system):
    """Split distance constraints into rigid 3-site waters (SETTLE) and X-H star clusters (SHAKE)."""
    n = system.getNumParticles()
    adj = [[] for _ in range(n)]
    for (i, j, d) in system.constraints:
        adj[i].append((j, d))
        adj[j].append((i, d))
    seen = [False] * n
    settle, shake, shake_d = [], [], []
    for a in range(n):
        if seen[a] or not adj[a]:
            continue
        comp, stack = [], [a]
        seen[a] = True
        while stack:
            u = stack.pop()
            comp.append(u)
            for (w, _) in adj[u]:
                if not seen[w]:
                    seen[w] = True
                    stack.append(w)
        comp.sort()
        ncons = sum(len(adj[u]) for u in comp) // 2
        if len(comp) == 3 and ncons == 3:
            # triangle: the heavy atom is the one with the largest mass
            o = max(comp, key=lambda u: system.masses[u])
            hs = [u for u in comp if u != o]
            d_oh = [d for (w, d) in adj[o] if w == hs[0]][0]
            d_oh2 = [d for (w, d) in adj[o] if w == hs[1]][0]
            d_hh = [d for (w, d) in adj[hs[0]] if w == hs[1]][0]
            if abs(d_oh - d_oh2) > 1e-9 or abs(system.masses[hs[0]] - system.masses[hs[1]]) > 1e-9:
                raise NotImplementedError('asymmetric rigid triangle constraints are not supported')
            settle.append((o, hs[0], hs[1], d_oh, d_hh))
        else:
            centre = max(comp, key=lambda u: len(adj[u]))
            if ncons != len(comp) - 1 or len(adj[centre]) != ncons or len(comp) > 4:
                raise NotImplementedError('constraint topology must be rigid water or X-H(1..3) star clusters')
            hs = [w for (w, _) in adj[centre]]
            ds = [d for (_, d) in adj[centre]]
            shake.append([centre] + hs + [-1] * (3 - len(hs)))
            shake_d.append(ds + [0.0] * (3 - len(ds)))
    return settle, shake, shake_d


def system_to_desc(system, box=None, ewald_split=None, min_edge=None):
    """Flatten a System into the arrays of remd_system_desc (include/remd_hip.h).

    ewald_split: None / 'reference' = OpenMM's rule (alpha and mesh from the NonbondedForce cutoff); 'auto' = the Coulomb range
    of ``rebalanced_coulomb_cutoff``; a number = that Coulomb range in nm.  With a range beyond the cutoff the descriptor
    carries ``coulomb_cutoff`` (-> remd_set_coulomb_cutoff) and the alpha / mesh that the same tolerance rule gives for it.
    """
    if ewald_split is None:
        ewald_split = os.environ.get('REMD_TOOLS_EWALD_SPLIT')        # diagnostic tools only (tools/*.py build descriptors directly)
    n = system.getNumParticles()
    d = dict(n_atoms=n, mass=np.array(system.masses, dtype=np.float64))
    d.update(n_ext=0, ext_atoms=np.zeros(0, np.int32), ext_K=0.0, ext_x0=0.0, ext_U0=0.0)
    bonds, angles, torsions = [], [], []
    nb = None
    gb = None
    cmm = 0
    # force groups of (external, bonds, angles, torsions, nonbonded direct, PME reciprocal): remd_set_force_groups
    fg = [0, 0, 0, 0, 0, 0]
    for f in system.forces:
        g = f.getForceGroup() if hasattr(f, 'getForceGroup') else 0
        if isinstance(f, CustomExternalForce):
            fg[0] = g
        elif isinstance(f, HarmonicBondForce):
            fg[1] = g
        elif isinstance(f, HarmonicAngleForce):
            fg[2] = g
        elif isinstance(f, PeriodicTorsionForce):
            fg[3] = g
        elif isinstance(f, NonbondedForce):
            fg[4] = g
            fg[5] = g if f.getReciprocalSpaceForceGroup() < 0 else f.getReciprocalSpaceForceGroup()
    d['force_groups'] = np.array(fg, dtype=np.int32)
    for f in system.forces:
        if isinstance(f, CustomExternalForce):
            d['n_ext'] = len(f.particles)
            d['ext_atoms'] = np.array(f.particles, dtype=np.int32)
            d['ext_K'] = f.globals['testsystems_HarmonicOscillator_K']
            d['ext_x0'] = f.globals['testsystems_HarmonicOscillator_x0']
            d['ext_U0'] = f.globals['testsystems_HarmonicOscillator_U0']
        elif isinstance(f, HarmonicBondForce):
            bonds += f.bonds
        elif isinstance(f, HarmonicAngleForce):
            angles += f.angles
        elif isinstance(f, PeriodicTorsionForce):
            torsions += f.torsions
        elif isinstance(f, NonbondedForce):
            if nb is not None:
                raise NotImplementedError('more than one NonbondedForce')
            nb = f
        elif isinstance(f, CMMotionRemover):
            cmm = f.frequency
        elif isinstance(f, GBSAOBCForce):
            gb = f
        else:
            raise NotImplementedError('unsupported force %r' % type(f).__name__)
    d['bond_atoms'] = np.array([b[:2] for b in bonds], dtype=np.int32).reshape(-1, 2)
    d['bond_params'] = np.array([b[2:] for b in bonds], dtype=np.float64).reshape(-1, 2)
    d['angle_atoms'] = np.array([a[:3] for a in angles], dtype=np.int32).reshape(-1, 3)
    d['angle_params'] = np.array([a[3:] for a in angles], dtype=np.float64).reshape(-1, 2)
    d['torsion_atoms'] = np.array([t[:4] for t in torsions], dtype=np.int32).reshape(-1, 4)
    d['torsion_params'] = np.array([t[4:] for t in torsions], dtype=np.float64).reshape(-1, 3)
    d.update(nb_method=0, cutoff=0.0, switch_distance=-1.0, rf_dielectric=78.3, ewald_alpha=0.0,
             pme_grid=np.zeros(3, np.int32), use_dispersion_correction=0,
             charge=np.zeros(n), sigma=np.zeros(n), epsilon=np.zeros(n),
             exception_atoms=np.zeros((0, 2), np.int32), exception_params=np.zeros((0, 3)))
    if nb is not None:
        if nb.getNumParticles() != n:
            raise ValueError('NonbondedForce has %d particles, system has %d' % (nb.getNumParticles(), n))
        method = nb.getNonbondedMethod()
        if method == NonbondedForce.CutoffPeriodic:
            d['nb_method'] = 1
        elif method == NonbondedForce.PME:
            d['nb_method'] = 2
        elif method == NonbondedForce.NoCutoff:
            d['nb_method'] = 3                      # REMD_NB_NOCUTOFF: every pair, no box (the vacuum test systems; csrc/nocutoff.hip)
        else:
            raise NotImplementedError('nonbonded method %d (only NoCutoff, CutoffPeriodic and PME are supported)' % method)
        nocut = d['nb_method'] == 3                 # OpenMM ignores cutoff, switching function and dispersion correction without a cutoff
        d['cutoff'] = 0.0 if nocut else nb.getCutoffDistance()
        d['switch_distance'] = nb.getSwitchingDistance() if (nb.getUseSwitchingFunction() and not nocut) else -1.0
        d['rf_dielectric'] = nb.getReactionFieldDielectric()
        d['use_dispersion_correction'] = int(nb.getUseDispersionCorrection() and not nocut)
        p = np.array(nb.particles, dtype=np.float64).reshape(-1, 3)
        d['charge'], d['sigma'], d['epsilon'] = p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy()
        d['exception_atoms'] = np.array([e[:2] for e in nb.exceptions], dtype=np.int32).reshape(-1, 2)
        d['exception_params'] = np.array([e[2:] for e in nb.exceptions], dtype=np.float64).reshape(-1, 3)
        if d['nb_method'] == 2:
            if box is None:
                box = np.diag(system.getDefaultPeriodicBoxVectors())
            if nb._pme_params is not None:
                d['ewald_alpha'] = nb._pme_params[0]
                d['pme_grid'] = np.array(nb._pme_params[1:], dtype=np.int32)
            else:
                tol = nb.getEwaldErrorTolerance()
                rcc = d['cutoff']
                if ewald_split == 'auto':
                    rcc = rebalanced_coulomb_cutoff(d['cutoff'], tol, box, min_edge=min_edge)
                elif ewald_split not in (None, 'reference'):
                    rcc = float(ewald_split)
                    if rcc < d['cutoff']:
                        raise ValueError('ewald_split: the Coulomb range cannot be shorter than the nonbonded cutoff')
                alpha, grid = ewald_parameters(rcc, tol, box)
                d['ewald_alpha'] = alpha
                d['pme_grid'] = np.array(grid, dtype=np.int32)
                if rcc > d['cutoff']:
                    d['coulomb_cutoff'] = rcc
    settle, shake, shake_d = _classify_constraints(system)
    d['settle_atoms'] = np.array([s[:3] for s in settle], dtype=np.int32).reshape(-1, 3)
    d['settle_dOH'] = settle[0][3] if settle else 0.0
    d['settle_dHH'] = settle[0][4] if settle else 0.0
    for s in settle:
        if abs(s[3] - d['settle_dOH']) > 1e-9 or abs(s[4] - d['settle_dHH']) > 1e-9:
            raise NotImplementedError('all rigid waters must share one geometry')
    d['shake_atoms'] = np.array(shake, dtype=np.int32).reshape(-1, 4)
    d['shake_dist'] = np.array(shake_d, dtype=np.float64).reshape(-1, 3)
    d['cmm_frequency'] = cmm
    region = getattr(system, 'alchemical_region', None)
    if region is not None:
        d['alch_atoms'] = np.array(sorted(region.alchemical_atoms), dtype=np.int32)
        d['softcore'] = (region.softcore_alpha, region.softcore_a, region.softcore_b, region.softcore_c)
        d['annihilate_sterics'] = bool(region.annihilate_sterics)          # -> remd_set_alchemical_options
    else:
        d['alch_atoms'] = np.zeros(0, np.int32)
        d['softcore'] = (0.5, 1.0, 1.0, 6.0)
        d['annihilate_sterics'] = False
    if gb is not None:
        # implicit solvent: remd_set_gbsa (csrc/gbsa.hip).  The surface term is the ACE one with OpenMM's default energy (28.3919551 = 4 pi x
        # 2.25936 kJ/mol/nm^2 in the factory's expression, alchemy.py:2207); 0 switches it off
        if d['nb_method'] != 3 or gb.getNonbondedMethod() != GBSAOBCForce.NoCutoff:
            raise NotImplementedError('GBSAOBCForce with a cutoff (only NoCutoff implicit-solvent systems are supported)')
        if gb.getNumParticles() != n:
            raise ValueError('GBSAOBCForce has %d particles, system has %d' % (gb.getNumParticles(), n))
        if gb.getSurfaceAreaEnergy() not in (0.0, 2.25936):
            raise NotImplementedError('GBSAOBCForce surface area energy %r (OpenMM\'s default 2.25936 kJ/mol/nm^2 or 0)' % gb.getSurfaceAreaEnergy())
        gp = np.array(gb.particles, dtype=np.float64).reshape(-1, 3)
        alch = np.zeros(n, dtype=np.int32)
        regions = getattr(system, 'alchemical_regions', None)
        if regions is not None:                       # the factory's alchemical GBSA: one region (alchemy.py:2168-2171)
            alch[regions[0].alchemical_atoms] = 1
        d['gbsa'] = dict(charge=gp[:, 0].copy(), radius=gp[:, 1].copy(), scale=gp[:, 2].copy(), alchemical=alch,
                         solute_dielectric=gb.getSoluteDielectric(), solvent_dielectric=gb.getSolventDielectric(),
                         surface_area=int(gb.getSurfaceAreaEnergy() != 0.0))
    if getattr(system, 'rf_unshifted_switch_width', None) is not None and d['nb_method'] == 1:
        # the reaction field as the alchemical factory re-writes it for the WHOLE system (alchemical_rf_treatment='switched'): remd_set_reaction_field
        d['rf_unshifted_switch_width'] = float(system.rf_unshifted_switch_width)
    if getattr(system, 'alchemical_regions', None) is not None:
        # general regions: this descriptor is the NonbondedForce the factory leaves behind, the custom forces follow through
        # remd_set_alchemical_regions (alchemy.AbsoluteAlchemicalFactory._region_terms)
        d['alch_regions'] = dict(system.alchemical_region_terms)
    return d


def from_openmm(omm_system):
    """Convert a real ``openmm.System`` (only where OpenMM is importable; not exercised in CI here)."""
    import openmm
    from openmm import unit as u
    s = System()
    for i in range(omm_system.getNumParticles()):
        s.addParticle(omm_system.getParticleMass(i).value_in_unit(u.amu))
    a, b, c = omm_system.getDefaultPeriodicBoxVectors()
    s.setDefaultPeriodicBoxVectors(*[v.value_in_unit(u.nanometer) for v in (a, b, c)])
    for i in range(omm_system.getNumConstraints()):
        p, q, dist = omm_system.getConstraintParameters(i)
        s.addConstraint(p, q, dist.value_in_unit(u.nanometer))
    for f in omm_system.getForces():
        if isinstance(f, openmm.HarmonicBondForce):
            g = HarmonicBondForce()
            for k in range(f.getNumBonds()):
                p, q, r0, kk = f.getBondParameters(k)
                g.addBond(p, q, r0.value_in_unit(u.nanometer), kk.value_in_unit(u.kilojoule_per_mole / u.nanometer ** 2))
        elif isinstance(f, openmm.HarmonicAngleForce):
            g = HarmonicAngleForce()
            for k in range(f.getNumAngles()):
                p, q, r, th, kk = f.getAngleParameters(k)
                g.addAngle(p, q, r, th.value_in_unit(u.radian), kk.value_in_unit(u.kilojoule_per_mole / u.radian ** 2))
        elif isinstance(f, openmm.PeriodicTorsionForce):
            g = PeriodicTorsionForce()
            for k in range(f.getNumTorsions()):
                p, q, r, t, per, ph, kk = f.getTorsionParameters(k)
                g.addTorsion(p, q, r, t, per, ph.value_in_unit(u.radian), kk.value_in_unit(u.kilojoule_per_mole))
        elif isinstance(f, openmm.NonbondedForce):
            g = NonbondedForce()
            for k in range(f.getNumParticles()):
                q, sig, eps = f.getParticleParameters(k)
                g.addParticle(q.value_in_unit(u.elementary_charge), sig.value_in_unit(u.nanometer),
                              eps.value_in_unit(u.kilojoule_per_mole))
            for k in range(f.getNumExceptions()):
                p, q, qq, sig, eps = f.getExceptionParameters(k)
                g.addException(p, q, qq.value_in_unit(u.elementary_charge ** 2), sig.value_in_unit(u.nanometer),
                               eps.value_in_unit(u.kilojoule_per_mole))
            g.setNonbondedMethod(f.getNonbondedMethod())
            g.setCutoffDistance(f.getCutoffDistance().value_in_unit(u.nanometer))
            g.setUseSwitchingFunction(f.getUseSwitchingFunction())
            g.setSwitchingDistance(f.getSwitchingDistance().value_in_unit(u.nanometer))
            g.setUseDispersionCorrection(f.getUseDispersionCorrection())
            g.setReactionFieldDielectric(f.getReactionFieldDielectric())
            g.setEwaldErrorTolerance(f.getEwaldErrorTolerance())
        elif isinstance(f, openmm.CMMotionRemover):
            g = CMMotionRemover(f.getFrequency())
        else:
            raise NotImplementedError('unsupported OpenMM force %s' % type(f).__name__)
        s.addForce(g)
    return s
