"""Alchemical regions and the lambda-dependent pieces of the energy that are host set-up.

Mirrors the parts of openmmtools/alchemy/alchemy.py the benchmark configs use:
  AlchemicalRegion defaults           :417-427 (softcore_alpha 0.5, a = b = 1, c = 6, annihilate_sterics False)
  AbsoluteAlchemicalFactory           :626-635, create_alchemical_system :637-754
  force split                         :1052-1083, :1539-2038: alchemical atoms lose their LJ in the
                                      NonbondedForce (eps = 0); alchemical/non-alchemical pairs go to a
                                      lambda_sterics-controlled soft-core CustomNonbondedForce (:1383-1388);
                                      alchemical/alchemical pairs keep full LJ (lambda fixed to 1, :1771-1775)
  dispersion correction               disable_alchemical_dispersion_correction=False (:630) => the custom
                                      forces use OpenMM's long-range correction, which depends on lambda.
The device evaluates the pair sums (csrc/forces.hip: nonbonded_kernel, alch_ukl_kernel); this module only
marks the region on the System and computes the per-state long-range-correction constants that
MultiStateSampler hands to remd_set_states(energy_const).
"""
import copy
import math
import numpy as np
from scipy import integrate

from .states import AlchemicalState, AlchemicalStateError          # alchemy.py:60-62, 90-410: users reach them as alchemy.AlchemicalState


class AlchemicalRegion:
    def __init__(self, alchemical_atoms=None, annihilate_electrostatics=True, annihilate_sterics=False,
                 softcore_alpha=0.5, softcore_a=1, softcore_b=1, softcore_c=6, softcore_beta=0.0,
                 softcore_d=1, softcore_e=1, softcore_f=2, name=None):
        if not alchemical_atoms:
            raise ValueError('The AlchemicalRegion is empty.')                      # alchemy.py:899-900 (raised there when the region is resolved)
        if (softcore_beta, softcore_d, softcore_e) != (0.0, 1, 1):
            raise NotImplementedError('softcore electrostatics (only the exact PME treatment is implemented)')
        self.alchemical_atoms = sorted(int(a) for a in alchemical_atoms)
        self.annihilate_electrostatics = annihilate_electrostatics
        self.annihilate_sterics = annihilate_sterics
        self.softcore_alpha, self.softcore_a, self.softcore_b, self.softcore_c = (
            float(softcore_alpha), float(softcore_a), float(softcore_b), float(softcore_c))
        self.name = name


class AbsoluteAlchemicalFactory:
    def __init__(self, consistent_exceptions=False, switch_width=0.1, alchemical_pme_treatment='exact',
                 alchemical_rf_treatment='switched', disable_alchemical_dispersion_correction=False,
                 split_alchemical_forces=True):
        if alchemical_pme_treatment not in ('exact', 'direct-space', 'coulomb'):
            raise ValueError(f"Unknown alchemical_pme_treatment scheme '{alchemical_pme_treatment}'")     # alchemy.py:1455
        if alchemical_rf_treatment not in ('switched', 'shifted'):
            raise ValueError(f"Unknown alchemical_rf_treatment scheme '{alchemical_rf_treatment}'")       # alchemy.py:1501
        if alchemical_pme_treatment != 'exact':
            raise NotImplementedError("only alchemical_pme_treatment='exact' (the reference default, alchemy.py:628)")
        self.disable_alchemical_dispersion_correction = disable_alchemical_dispersion_correction

    def create_alchemical_system(self, reference_system, alchemical_regions, alchemical_regions_interactions=frozenset()):
        """alchemy.py:637-664.  alchemical_regions_interactions names pairs of regions that interact through their own lambdas: it
        only has a meaning with several regions, which are not built here."""
        if alchemical_regions_interactions != frozenset():
            raise NotImplementedError('interactions between several alchemical regions (alchemy.py:661-664)')
        if isinstance(alchemical_regions, (list, tuple)):
            if len(alchemical_regions) != 1:
                raise NotImplementedError('multiple alchemical regions')
            alchemical_regions = alchemical_regions[0]
        system = copy.deepcopy(reference_system)
        n = system.getNumParticles()
        if max(alchemical_regions.alchemical_atoms) >= n:
            raise ValueError('alchemical atom index out of range')
        system.alchemical_region = alchemical_regions
        system.alchemical_lrc = not self.disable_alchemical_dispersion_correction
        return system


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


def alchemical_long_range_constants(system, nonbonded_force, lambdas_sterics, volume):
    """Per-state long-range correction (kJ/mol) of the two sterics CustomNonbondedForces.

    OpenMM CustomNonbondedForce convention: E = 2 pi N^2 / V * sum_classpairs count * I / (N (N+1)/2),
    with count restricted to the force's interaction group: alchemical x non-alchemical atoms for the
    lambda-controlled force, alchemical pairs (lambda = 1) for the other.
    """
    region = system.alchemical_region
    if region is None or not getattr(system, 'alchemical_lrc', True) or not nonbonded_force.getUseDispersionCorrection():
        return np.zeros(len(lambdas_sterics))
    n = system.getNumParticles()
    rc = nonbonded_force.getCutoffDistance()
    rs = nonbonded_force.getSwitchingDistance() if nonbonded_force.getUseSwitchingFunction() else None
    alch = set(region.alchemical_atoms)
    cls_a, cls_n = {}, {}
    for i, (q, s, e) in enumerate(nonbonded_force.particles):
        d = cls_a if i in alch else cls_n
        d[(s, e)] = d.get((s, e), 0) + 1
    norm = 2.0 * math.pi * n * n / (n * (n + 1) / 2.0) / volume
    out = np.zeros(len(lambdas_sterics))
    for k, lam in enumerate(lambdas_sterics):
        tot = 0.0
        for (sa, ea), na in cls_a.items():
            for (sn, en), nn in cls_n.items():
                eps = math.sqrt(ea * en)
                if eps > 0:
                    tot += na * nn * _tail_integral(0.5 * (sa + sn), eps, lam, region, rc, rs)
        # alchemical/alchemical force: lambda fixed to 1 unless annihilate_sterics (alchemy.py:1767-1779)
        lam_aa = lam if region.annihilate_sterics else 1.0
        keys = list(cls_a.items())
        for x in range(len(keys)):
            for y in range(x, len(keys)):
                (s1, e1), n1 = keys[x]
                (s2, e2), n2 = keys[y]
                count = n1 * (n1 - 1) / 2.0 if x == y else n1 * n2
                eps = math.sqrt(e1 * e2)
                if eps > 0 and count > 0:
                    tot += count * _tail_integral(0.5 * (s1 + s2), eps, lam_aa, region, rc, rs)
        out[k] = norm * tot
    return out
