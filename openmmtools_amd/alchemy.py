"""Alchemical regions and the lambda-dependent pieces of the energy that are host set-up.

Mirrors the parts of openmmtools/alchemy/alchemy.py that define WHAT the alchemical Hamiltonian is:
  AlchemicalRegion defaults           :417-427 (softcore_alpha 0.5, a = b = 1, c = 6, beta 0, d = e = 1, f = 2, annihilate_sterics False)
  AbsoluteAlchemicalFactory           :626-635, create_alchemical_system :637-754
  force split                         :1052-1083, :1539-2038: alchemical atoms lose their LJ in the
                                      NonbondedForce (eps = 0); alchemical/non-alchemical pairs go to a
                                      lambda_sterics-controlled soft-core CustomNonbondedForce (:1383-1388);
                                      alchemical/alchemical pairs keep full LJ (lambda fixed to 1, :1771-1775)
  dispersion correction               disable_alchemical_dispersion_correction=False (:630) => the custom
                                      forces use OpenMM's long-range correction, which depends on lambda.

Two device paths serve it:
  * ONE region under the exact PME treatment (or without alchemical charges), softcore_c = 6 -- the benchmark configs -- is folded into
    the pair kernels (csrc/forces.hip: soft-core flag per atom, charges scaled by lambda_electrostatics, alch_ukl_kernel): the System
    is marked with ``alchemical_region``;
  * everything else the factory builds for a NonbondedForce -- several named regions with their own lambdas (:1360-1377),
    alchemical_regions_interactions (:661-664, 1684-1690), the 'direct-space' / 'coulomb' PME treatments and the reaction-field
    treatments with soft-core electrostatics (:1392-1537), any softcore exponents, decoupled electrostatics -- runs as the factory's own
    force split: the System keeps the NonbondedForce the factory leaves behind (alchemical charges and epsilons zeroed) and carries
    ``alchemical_regions`` + ``alchemical_region_terms`` (the custom forces' parameters) for csrc/alch_regions.hip
    (remd_set_alchemical_regions).
Under the exact PME treatment that path scales every region's charges by its own lambda_electrostatics inside the whole Ewald sum
(remd_alch_regions_desc.exact_pme).  It also carries the softened bonds / angles / torsions of a region (:1115-1354; lambda_bonds ...),
the vacuum systems (NonbondedForce.NoCutoff: csrc/nocutoff.hip) and, with a GBSAOBCForce in the System, the alchemical GBSA of
:2144-2225 (csrc/gbsa.hip; one region, as in the reference).  Not built: AMOEBA, GB models other than OBC2, GBSA with a cutoff.
This module also computes the per-state long-range-correction constants that MultiStateSampler hands to remd_set_states(energy_const).
"""
import copy
import math
import numpy as np
from scipy import integrate

from .states import AlchemicalState, AlchemicalStateError          # alchemy.py:60-62, 90-410: users reach them as alchemy.AlchemicalState


class AlchemicalRegion:
    """alchemy.py:416-600 (a namedtuple there; same fields, same order, same defaults).  alchemical_bonds / angles / torsions: None (no
    softened bonded terms), True (every bond / angle / proper torsion that involves an alchemical atom, :940-1050) or the indices of the
    terms in the System's HarmonicBondForce / HarmonicAngleForce / PeriodicTorsionForce."""

    def __init__(self, alchemical_atoms=None, alchemical_bonds=None, alchemical_angles=None, alchemical_torsions=None,
                 annihilate_electrostatics=True, annihilate_sterics=False,
                 softcore_alpha=0.5, softcore_a=1, softcore_b=1, softcore_c=6, softcore_beta=0.0,
                 softcore_d=1, softcore_e=1, softcore_f=2, name=None):
        if not alchemical_atoms:
            raise ValueError('The AlchemicalRegion is empty.')                      # alchemy.py:899-900 (raised there when the region is resolved)
        self.alchemical_atoms = sorted(int(a) for a in alchemical_atoms)
        self.alchemical_bonds, self.alchemical_angles, self.alchemical_torsions = (
            (t if (t is None or t is True or t is False) else sorted(int(k) for k in t)) for t in (alchemical_bonds, alchemical_angles, alchemical_torsions))
        self.annihilate_electrostatics = bool(annihilate_electrostatics)
        self.annihilate_sterics = bool(annihilate_sterics)
        self.softcore_alpha, self.softcore_a, self.softcore_b, self.softcore_c = (
            float(softcore_alpha), float(softcore_a), float(softcore_b), float(softcore_c))
        self.softcore_beta, self.softcore_d, self.softcore_e, self.softcore_f = (
            float(softcore_beta), float(softcore_d), float(softcore_e), float(softcore_f))
        self.name = name

    @property
    def softcore(self):
        """alpha, beta, a, b, c, d, e, f: the order of remd_alch_regions_desc.softcore"""
        return (self.softcore_alpha, self.softcore_beta, self.softcore_a, self.softcore_b, self.softcore_c,
                self.softcore_d, self.softcore_e, self.softcore_f)


class AbsoluteAlchemicalFactory:
    def __init__(self, consistent_exceptions=False, switch_width=0.1, alchemical_pme_treatment='exact',
                 alchemical_rf_treatment='switched', disable_alchemical_dispersion_correction=False,
                 split_alchemical_forces=True):
        if alchemical_pme_treatment not in ('exact', 'direct-space', 'coulomb'):
            raise ValueError(f"Unknown alchemical_pme_treatment scheme '{alchemical_pme_treatment}'")     # alchemy.py:1455
        if alchemical_rf_treatment not in ('switched', 'shifted'):
            raise ValueError(f"Unknown alchemical_rf_treatment scheme '{alchemical_rf_treatment}'")       # alchemy.py:1501
        self.consistent_exceptions = bool(consistent_exceptions)
        self.switch_width = float(switch_width)
        self.alchemical_pme_treatment = alchemical_pme_treatment
        self.alchemical_rf_treatment = alchemical_rf_treatment
        self.disable_alchemical_dispersion_correction = disable_alchemical_dispersion_correction
        self.split_alchemical_forces = split_alchemical_forces

    def create_alchemical_system(self, reference_system, alchemical_regions, alchemical_regions_interactions=frozenset()):
        """alchemy.py:637-754."""
        from .system import NonbondedForce
        if isinstance(alchemical_regions, AlchemicalRegion):
            alchemical_regions = [alchemical_regions]
        regions = list(alchemical_regions)
        interactions = sorted(set((min(int(a), int(b)), max(int(a), int(b))) for a, b in alchemical_regions_interactions))
        if interactions and len(regions) < 2:
            raise ValueError('alchemical_regions_interactions names pairs of regions: there is only one')
        for a, b in interactions:
            if a == b or not (0 <= a < len(regions) and 0 <= b < len(regions)):
                raise ValueError('alchemical_regions_interactions: (%d, %d) is not a pair of regions' % (a, b))
        names = [r.name for r in regions]
        if len(regions) > 1 and (None in names or len(set(names)) != len(names)):
            raise ValueError('several alchemical regions need distinct names')                          # alchemy.py:666-672
        system = copy.deepcopy(reference_system)
        n = system.getNumParticles()
        seen = set()
        for r in regions:
            if max(r.alchemical_atoms) >= n:
                raise ValueError('alchemical atom index out of range')
            if seen & set(r.alchemical_atoms):
                raise ValueError('alchemical regions overlap')
            seen |= set(r.alchemical_atoms)
        system.alchemical_lrc = not self.disable_alchemical_dispersion_correction
        nbs = [f for f in system.getForces() if isinstance(f, NonbondedForce)]
        nb = nbs[0] if nbs else None
        if nb is not None and nb.getNonbondedMethod() == NonbondedForce.CutoffPeriodic and self.alchemical_rf_treatment == 'switched':
            # alchemy.py:744-749: the factory then replaces the reaction field of the WHOLE system by an unshifted, switched one
            # (forcefactories.replace_reaction_field :76-84, forces.UnshiftedReactionFieldForce): remd_set_reaction_field
            system.rf_unshifted_switch_width = self.switch_width
        charged = nb is not None and any(nb.particles[i][0] != 0.0 for i in seen) or \
            (nb is not None and any(e[2] != 0.0 and (e[0] in seen or e[1] in seen) for e in nb.exceptions))
        is_pme = nb is not None and nb.getNonbondedMethod() == NonbondedForce.PME
        exact = is_pme and self.alchemical_pme_treatment == 'exact'
        if exact:                                                                                         # alchemy.py:1616-1625
            err = ' not supported with exact treatment of Ewald electrostatics.'
            for r in regions:
                if not r.annihilate_electrostatics:
                    raise ValueError('Decoupled electrostatics is' + err)
                if self.consistent_exceptions:
                    raise ValueError('Consistent exceptions are' + err)
                if (r.softcore_beta, r.softcore_d, r.softcore_e) != (0, 1, 1):
                    raise ValueError('Softcore electrostatics is' + err)
        from .system import GBSAOBCForce
        has_gb = any(isinstance(f, GBSAOBCForce) for f in system.getForces())
        if has_gb and len(regions) > 1:
            raise NotImplementedError('Multiple regions does not work with GBSAOBCForce')               # alchemy.py:2168-2169
        r0 = regions[0]
        bonded = self._softened_bonded_terms(system, regions, interactions)           # takes them out of the System's bonded forces
        fast = (len(regions) == 1 and r0.softcore_c == 6.0 and (exact or not charged) and bonded is None and
                nb is not None and nb.getNonbondedMethod() != NonbondedForce.NoCutoff)      # (that path lives in the cutoff-based pair kernels)
        if fast:
            # the pair kernels' own path: one region, charges scaled inside the Ewald sum or none to scale
            system.alchemical_region = r0
            system.alchemical_regions = None
            return system
        if nb is None:
            raise ValueError('alchemical regions need a NonbondedForce')
        system.alchemical_region = None
        system.alchemical_regions = regions
        # alchemical_regions_interactions, as the reference's loop EXECUTES them (alchemy.py:1693, 1886-1911): the forces of a pair of
        # regions are built after both regions' single turns, from the NonbondedForce in which both regions' atoms (and their
        # exceptions) have been zeroed already -- their particle tables carry charge 0 and epsilon 0, so they evaluate to zero; only
        # under the exact PME treatment does the pair matter (no exclusions between the two regions, :1663-1672).  The engine's class
        # of interacting regions (product of the lambdas, remd_alch_regions_desc.interactions) is therefore not used by this factory;
        # the pairs are recorded for the store writer (_alchemical_xml.py), which emits those forces as the reference would.
        system.alchemical_regions_interactions = interactions
        system.alchemical_factory_options = dict(alchemical_pme_treatment=self.alchemical_pme_treatment, alchemical_rf_treatment=self.alchemical_rf_treatment,
                                                 switch_width=self.switch_width, consistent_exceptions=self.consistent_exceptions)
        system.alchemical_region_terms = self._region_terms(nb, regions, charged, exact, interactions)
        if bonded is not None:
            system.alchemical_region_terms.update(bonded)
        system.alchemical_region_terms['consistent_exceptions'] = int(self.consistent_exceptions)          # alchemy.py:1456-1461
        return system

    # ---- softened bonds / angles / torsions (alchemy.py:940-1050 what True means, :1115-1354 the forces) -------------------------
    @staticmethod
    def _softened_bonded_terms(system, regions, interactions):
        from .system import HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce
        if all(not getattr(r, 'alchemical_' + k) for r in regions for k in ('bonds', 'angles', 'torsions')):
            return None
        n = system.getNumParticles()
        bonded_to = [set() for _ in range(n)]                               # _tabulate_bonds: bonds and constraints
        for f in system.getForces():
            if isinstance(f, HarmonicBondForce):
                for b in f.bonds:
                    bonded_to[b[0]].add(b[1]); bonded_to[b[1]].add(b[0])
        for k in range(system.getNumConstraints()):
            i, j, _ = system.getConstraintParameters(k)
            bonded_to[i].add(j); bonded_to[j].add(i)
        region_of = {}
        for g, r in enumerate(regions):
            for a in r.alchemical_atoms:
                region_of[a] = g
        together = set(interactions)

        def straddling(atoms):                                             # _are_straddling_noninteracting_regions (:1086-1112)
            gs = sorted({region_of[a] for a in atoms if a in region_of})
            return len(gs) > 1 and (gs[0], gs[1]) not in together
        out = {}
        for kind, cls, attr, width in (('bond', HarmonicBondForce, 'bonds', 2), ('angle', HarmonicAngleForce, 'angles', 3), ('torsion', PeriodicTorsionForce, 'torsions', 4)):
            forces = [f for f in system.getForces() if isinstance(f, cls)]
            chosen = {}
            for g, r in enumerate(regions):
                want = getattr(r, 'alchemical_' + kind + 's')
                if not want:
                    continue
                if not forces:
                    raise ValueError('alchemical_%ss without a %s in the system' % (kind, cls.__name__))
                terms = getattr(forces[-1], attr)
                if want is True:
                    A = set(r.alchemical_atoms)
                    want = [k for k, t in enumerate(terms) if A & set(t[:width]) and
                            (kind != 'torsion' or all(t[q + 1] in bonded_to[t[q]] for q in range(3)))]           # proper torsions only (:970-990)
                bad = [k for k in want if not 0 <= k < len(terms)]
                if bad:
                    raise ValueError('Indices {} in {} cannot be found in the system'.format(set(bad), 'alchemical_%ss' % kind))    # alchemy.py:886-889
                for k in want:
                    chosen[k] = g + 1                                   # (a term two regions claim: the later region's force holds it last)
            if not chosen:
                continue
            f = forces[-1]
            terms = getattr(f, attr)
            out[kind + '_atoms'] = np.array([terms[k][:width] for k in sorted(chosen)], dtype=np.int32).reshape(-1, width)
            out[kind + '_params'] = np.array([terms[k][width:] for k in sorted(chosen)], dtype=np.float64)
            out[kind + '_region'] = np.array([chosen[k] for k in sorted(chosen)], dtype=np.int32)
            out[kind + '_index'] = np.array(sorted(chosen), dtype=np.int32)
            # what stays in the plain force: not softened, and not connecting two regions that do not interact (:1156-1162)
            setattr(f, attr, [t for k, t in enumerate(terms) if k not in chosen and not straddling(t[:width])])
        return out or None

    # ---- the factory's split of a NonbondedForce (alchemy.py:1539-2038) ---------------------------------------------
    def _region_terms(self, nb, regions, charged, exact=False, interactions=()):
        """Zero the alchemical atoms in ``nb`` (what the factory leaves in the NonbondedForce, :1903-1911, 2001-2006) and return the
        parameters of the custom forces: the dict system_to_desc passes on as ``alch_regions`` (remd_alch_regions_desc)."""
        from .system import NonbondedForce
        n = nb.getNumParticles()
        region_of = np.zeros(n, dtype=np.int32)
        for g, r in enumerate(regions):
            region_of[r.alchemical_atoms] = g + 1
        p = np.array(nb.particles, dtype=np.float64).reshape(-1, 3)
        p[p[:, 1] == 0.0, 1] = 0.1                                       # sigma = 0 -> 1 A (:1638-1648)
        exc_atoms, exc_params = [], []
        together = set(interactions)
        for k, e in enumerate(nb.exceptions):
            i, j, qq, sig, eps = e
            if region_of[i] == 0 and region_of[j] == 0:
                continue
            if exact and charged and region_of[i] and region_of[j] and region_of[i] != region_of[j] and \
                    (min(region_of[i], region_of[j]) - 1, max(region_of[i], region_of[j]) - 1) not in together:
                # exact PME treatment: two regions that do not interact exclude each other BEFORE the loop -- addException(atom1, atom2,
                # 0.0, 1.0, 0.0, replace=True), alchemy.py:1663-1672 -- which wipes an exception between them
                nb.exceptions[k] = (i, j, 0.0, 1.0, 0.0)
                continue
            if sig == 0.0:
                sig = 0.1                                                # (:1650-1661)
            if qq != 0.0 or eps != 0.0:
                # (between two regions: the first region's turn takes it as "only one alchemical" and zeroes it, :1972-1976, 1992-2006;
                # the check of :1966-1969 comes after that and never sees a charge product)
                exc_atoms.append((i, j)); exc_params.append((qq, sig, eps))
            nb.exceptions[k] = (i, j, 0.0, sig, 0.0)
        for i in np.nonzero(region_of)[0]:
            q, sig, eps = nb.particles[i]
            nb.particles[i] = (0.0, sig if sig != 0.0 else 0.1, 0.0)
        rc = nb.getCutoffDistance()
        method = nb.getNonbondedMethod()
        terms = dict(region_of_atom=region_of, softcore=np.array([r.softcore for r in regions], dtype=np.float64),
                     annihilate=np.array([[r.annihilate_sterics, r.annihilate_electrostatics] for r in regions], dtype=np.int32),
                     interactions=np.zeros((0, 2), dtype=np.int32),
                     charge=p[:, 0].copy(), sigma=p[:, 1].copy(), epsilon=p[:, 2].copy(),
                     exception_atoms=np.array(exc_atoms, dtype=np.int32).reshape(-1, 2),
                     exception_params=np.array(exc_params, dtype=np.float64).reshape(-1, 3),
                     electrostatics=int(bool(charged)), elec_alpha=0.0, elec_krf=0.0, elec_crf=0.0, elec_switch_distance=-1.0, exact_pme=0)
        if exact and charged:
            # the exact PME treatment (alchemy.py:1663-1681, 1893-1899, 1978-1982): no electrostatic custom forces; `charge` and the
            # exceptions above are the parameter OFFSETS of the NonbondedForce (scaled by the region's lambda_electrostatics inside the
            # whole Ewald sum), regions that do not interact exclude each other, regions that do see each other's scaled charges
            terms['exact_pme'] = 1
            terms['electrostatics'] = 0
            terms['interactions'] = np.array([(a + 1, b + 1) for a, b in interactions], dtype=np.int32).reshape(-1, 2)
        elif charged:
            if method == NonbondedForce.PME:
                if self.alchemical_pme_treatment == 'direct-space':                                      # :1510-1537
                    alpha = nb._pme_params[0] if nb._pme_params is not None else 0.0
                    if alpha == 0.0:
                        alpha = math.sqrt(-math.log(2.0 * nb.getEwaldErrorTolerance())) / rc
                    terms['elec_alpha'] = float(alpha)
                else:                                                                                      # 'coulomb': switched plain Coulomb (:1449-1452, 1818-1821)
                    terms['elec_switch_distance'] = rc - self.switch_width
            elif method == NonbondedForce.CutoffPeriodic:                                                 # :1473-1508
                eps_s = nb.getReactionFieldDielectric()
                terms['elec_krf'] = rc ** -3 * (eps_s - 1.0) / (2.0 * eps_s + 1.0)
                if self.alchemical_rf_treatment == 'switched':
                    terms['elec_switch_distance'] = rc - self.switch_width
                else:
                    terms['elec_crf'] = (1.0 / rc) * 3.0 * eps_s / (2.0 * eps_s + 1.0)
            elif method != NonbondedForce.NoCutoff:                                                       # NoCutoff: plain l^d qq / r_eff, no switch (:1434-1447)
                raise NotImplementedError('nonbonded method %d (only NoCutoff, CutoffPeriodic and PME are supported)' % method)
        return terms


def region_lambda_of_class(kind, ls_a, ls_b, annihilate):
    """lambda of a class of pairs: (environment, a), (a, a) or (a, b) -- alchemy.py:1766-1779."""
    if kind == 0:
        return ls_a
    if kind == 1:
        return ls_a if annihilate else 1.0
    return ls_a * ls_b


def _softcore_energy(r, sigma, eps, lam, region):
    a, b, c, alpha = region.softcore_a, region.softcore_b, region.softcore_c, region.softcore_alpha
    reff = sigma * (alpha * (1.0 - lam) ** b + (r / sigma) ** c) ** (1.0 / c)       # alchemy.py:1388
    x = (sigma / reff) ** 6
    return lam ** a * 4.0 * eps * x * (x - 1.0)                                      # :1385-1386


def _tail_integral(sigma, eps, lam, region, rc, rs):
    """int_{rc}^inf U r^2 dr + int_{rs}^{rc} (1 - S) U r^2 dr for the soft-core pair potential."""
    f = lambda r: _softcore_energy(r, sigma, eps, lam, region) * r * r
    tail = integrate.quad(f, rc, np.inf, epsabs=0, epsrel=1e-11)[0]
    if rs is not None and 0 <= rs < rc:
        def g(r):
            x = (r - rs) / (rc - rs)
            S = 1.0 - 10.0 * x ** 3 + 15.0 * x ** 4 - 6.0 * x ** 5
            return (1.0 - S) * f(r)
        tail += integrate.quad(g, rs, rc, epsabs=0, epsrel=1e-11)[0]
    return tail


def _classes(sigma, epsilon, atoms):
    out = {}
    for i in atoms:
        out[(sigma[i], epsilon[i])] = out.get((sigma[i], epsilon[i]), 0) + 1
    return list(out.items())


def _group_sum(A, B, same, lam, region, rc, rs, cache):
    """sum over the pairs of an interaction group (set A x set B; ``same``: A is B, unordered pairs) of the tail integrals"""
    tot = 0.0
    for x, ((s1, e1), n1) in enumerate(A):
        for y, ((s2, e2), n2) in enumerate(B):
            if same and y < x:
                continue
            count = (n1 * (n1 - 1) / 2.0 if x == y else n1 * n2) if same else n1 * n2
            eps = math.sqrt(e1 * e2)
            if eps > 0 and count > 0:
                key = (0.5 * (s1 + s2), eps, lam, id(region))
                if key not in cache:
                    cache[key] = _tail_integral(0.5 * (s1 + s2), eps, lam, region, rc, rs)
                tot += count * cache[key]
    return tot


def alchemical_long_range_constants(system, nonbonded_force, lambdas_sterics, volume):
    """Per-state long-range correction (kJ/mol) of the sterics CustomNonbondedForces.

    OpenMM CustomNonbondedForce convention: E = 2 pi N^2 / V * sum_classpairs count * I / (N (N+1)/2),
    with count restricted to the force's interaction group: alchemical x non-alchemical atoms for the
    lambda-controlled force, alchemical pairs (lambda = 1) for the other.

    lambdas_sterics: [K] for a System with ``alchemical_region`` (one region); [K][n_regions] for ``alchemical_regions`` -- then every
    single region has its two forces and every pair of interacting regions one, controlled by the product of the two lambdas
    (alchemy.py:1766-1779, 1786-1789, 1916-1922).
    """
    regions = getattr(system, 'alchemical_regions', None)
    if regions is None:
        region = getattr(system, 'alchemical_region', None)
        if region is None:
            return np.zeros(len(lambdas_sterics))
        regions, interactions = [region], []
        lam = np.asarray(lambdas_sterics, dtype=np.float64).reshape(-1, 1)
        sigma = [q[1] for q in nonbonded_force.particles]
        epsilon = [q[2] for q in nonbonded_force.particles]
    else:
        terms = system.alchemical_region_terms
        interactions = [(int(a) - 1, int(b) - 1) for a, b in terms['interactions']]         # (none from this package's factory)
        lam = np.asarray(lambdas_sterics, dtype=np.float64).reshape(-1, len(regions))
        sigma, epsilon = list(terms['sigma']), list(terms['epsilon'])
    if not getattr(system, 'alchemical_lrc', True) or not nonbonded_force.getUseDispersionCorrection():
        return np.zeros(len(lam))
    n = system.getNumParticles()
    rc = nonbonded_force.getCutoffDistance()
    rs = nonbonded_force.getSwitchingDistance() if nonbonded_force.getUseSwitchingFunction() else None
    alch = set()
    for r in regions:
        alch |= set(r.alchemical_atoms)
    env = _classes(sigma, epsilon, [i for i in range(n) if i not in alch])
    per_region = [_classes(sigma, epsilon, r.alchemical_atoms) for r in regions]
    norm = 2.0 * math.pi * n * n / (n * (n + 1) / 2.0) / volume
    out = np.zeros(len(lam))
    cache = {}
    for k in range(len(lam)):
        tot = 0.0
        for g, region in enumerate(regions):
            tot += _group_sum(per_region[g], env, False, lam[k, g], region, rc, rs, cache)
            # alchemical/alchemical force: lambda fixed to 1 unless annihilate_sterics (alchemy.py:1767-1779)
            tot += _group_sum(per_region[g], per_region[g], True, lam[k, g] if region.annihilate_sterics else 1.0, region, rc, rs, cache)
        for a, b in interactions:
            tot += _group_sum(per_region[a], per_region[b], False, lam[k, a] * lam[k, b], regions[b], rc, rs, cache)
        out[k] = norm * tot
    return out
