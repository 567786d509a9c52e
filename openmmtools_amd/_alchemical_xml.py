"""The alchemically modified System as OpenMM XML: what the reference's AbsoluteAlchemicalFactory would have built, written
from and read back into this package's *marked* System (``system.alchemical_region``, alchemy.py here).

The reference stores a CompoundThermodynamicState's System as ``XmlSerializer.serialize`` of the factory's output
(states.py:1257-1280, multistatereporter.py:612-668); to put configs 2 and 4 into the reference's netCDF4 layout, the store
writer therefore needs that force set (alchemy/alchemy.py of the reference, defaults of :626-635: exact PME treatment,
split_alchemical_forces, no consistent exceptions, switched reaction field):

    order        forces the factory does not touch, in place; then the re-added bonded forces (and the NonbondedForce when
                 electrostatics is not handled by offsets) :711-741, 1052-1061; then by sorted lambda name :1075-1083 --
                 'lambda_electrostatics' (force group = lowest free) and 'lambda_sterics' (next free); the unshifted
                 reaction-field force last :744-749 (forcefactories.py:76-84, forces.py:1110-1196)
    NonbondedForce   alchemical atoms lose charge and epsilon, exceptions that touch them lose chargeprod and epsilon
                 :1903-1911, 2000-2006; Ewald methods: global parameter lambda_electrostatics + one particle offset (q) per
                 alchemical atom :1675-1680, 1896-1899 and one exception offset per charged exception :1978-1982;
                 sigma = 0 becomes 1 angstrom :1640-1661
    sterics      CustomNonbondedForce non-alchemical/alchemical (lambda_sterics, interaction group (N, A)) and
                 alchemical/alchemical (group (A, A); lambda fixed to 1 in the expression unless annihilate_sterics)
                 :1767-1797, 1913-1919, expressions :1355-1390; CustomBondForces for the Lennard-Jones part of exceptions
                 with one / two alchemical atoms :1836-1851, 1985-1998
    electrostatics   (non-Ewald methods only) the same four forces with the reaction-field expression :1392-1508, 1799-1830
    softcore_*   eight global parameters on every custom force :2009-2025

The XML element layout of CustomNonbondedForce / CustomBondForce / NonbondedForce offsets restates OpenMM's
CustomNonbondedForceProxy, CustomBondForceProxy and NonbondedForceProxy -- **external knowledge, unpinned**: neither OpenMM
nor any document it wrote with these forces exists in /root/reference or here (system_xml.py's plain forces are pinned by
the System inside the reference's alanine store).  What is pinned: write -> read returns the System description the engine
was given (tests/test_alchemical_store_cpu.py), so a store written here resumes here.

Outside what this package's factory builds (and so refused, with the name of what is missing): several regions, alchemical
bonds / angles / torsions, soft-core electrostatics, 'coulomb' / 'direct-space' PME treatments, charged systems under the
reaction-field methods (the engine evaluates OpenMM's shifted reaction field, the factory an unshifted switched one).
"""
import math
import re
import xml.etree.ElementTree as ET

from .constants import ONE_4PI_EPS0
from .system import NonbondedForce, HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce

_REMODELLED = (HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce, NonbondedForce)      # :711-726 have a creator
_SOFTCORE = ('softcore_alpha', 'softcore_beta', 'softcore_a', 'softcore_b', 'softcore_c', 'softcore_d', 'softcore_e', 'softcore_f')
_MIX_STERICS = 'epsilon = sqrt(epsilon1*epsilon2);sigma = 0.5*(sigma1 + sigma2);'
_MIX_ELECTROSTATICS = 'chargeprod = charge1*charge2;sigma = 0.5*(sigma1 + sigma2);'
_RF_HEAD = 'ONE_4PI_EPS0*chargeprod*(r^(-1) + k_rf*r^2);'


def _f(x):
    return repr(float(x))


def sterics_exception_expression(lam='lambda_sterics'):
    """:1374-1380 -- the soft-core Lennard-Jones of one pair with effective sigma / epsilon."""
    return ('U_sterics;'
            'U_sterics = ((%s)^softcore_a)*4*epsilon*x*(x-1.0);'
            'x = (sigma/reff_sterics)^6;'
            'reff_sterics = sigma*((softcore_alpha*(1.0-(%s))^softcore_b + (r/sigma)^softcore_c))^(1/softcore_c);' % (lam, lam))


def electrostatics_expressions(nb, lam='lambda_electrostatics', pme_treatment='direct-space', rf_treatment='switched', consistent_exceptions=False):
    """:1392-1471: (pair expression, exception expression) for NoCutoff, the reaction-field treatments (:1473-1508) and the
    'direct-space' (:1510-1537) / 'coulomb' PME treatments."""
    prefix = 'U_electrostatics;U_electrostatics=((%s)^softcore_d)*ONE_4PI_EPS0*chargeprod' % lam
    suffix = ('reff_electrostatics = sigma*((softcore_beta*(1.0-(%s))^softcore_e + (r/sigma)^softcore_f))^(1/softcore_f);'
              'ONE_4PI_EPS0 = %s;' % (lam, ONE_4PI_EPS0))
    coulomb = '/reff_electrostatics;'
    m = nb.getNonbondedMethod()
    if m == NonbondedForce.NoCutoff:
        method = coulomb
    elif m in (NonbondedForce.CutoffPeriodic, NonbondedForce.CutoffNonPeriodic):        # :1473-1508
        eps, rc = nb.getReactionFieldDielectric(), nb.getCutoffDistance()
        c_rf = 0.0 if rf_treatment == 'switched' else rc ** (-1) * ((3 * eps) / (2 * eps + 1))
        method = ('*(reff_electrostatics^(-1) + k_rf*reff_electrostatics^2 - c_rf);k_rf = %s;c_rf = %s;'
                  % (rc ** -3 * ((eps - 1) / (2 * eps + 1)), c_rf))
    elif pme_treatment == 'direct-space':                                                # :1510-1537
        alpha = nb._pme_params[0] if nb._pme_params is not None else 0.0
        if alpha == 0.0:
            alpha = (1.0 / nb.getCutoffDistance()) * math.sqrt(-math.log(2.0 * nb.getEwaldErrorTolerance()))
        method = '*erfc(alpha_ewald*reff_electrostatics)/reff_electrostatics;alpha_ewald = %s;' % alpha
    else:
        method = coulomb
    return prefix + method + suffix + _MIX_ELECTROSTATICS, prefix + (method if consistent_exceptions else coulomb) + suffix       # :1456-1461


def _emit_custom(forces, kind, group, energy, per_name, per_params, lam_globals, region, **attrs):
    e = ET.SubElement(forces, 'Force', dict(forceGroup=str(group), name=kind, type=kind, energy=energy, version='3', **attrs))
    b = ET.SubElement(e, per_name)
    for n in per_params:
        ET.SubElement(b, 'Parameter', dict(name=n))
    g = ET.SubElement(e, 'GlobalParameters')
    values = dict(softcore_alpha=region.softcore_alpha, softcore_beta=getattr(region, 'softcore_beta', 0.0), softcore_a=region.softcore_a,
                  softcore_b=region.softcore_b, softcore_c=region.softcore_c, softcore_d=getattr(region, 'softcore_d', 1.0),
                  softcore_e=getattr(region, 'softcore_e', 1.0), softcore_f=getattr(region, 'softcore_f', 2.0))
    for n in lam_globals:
        ET.SubElement(g, 'Parameter', dict(default=_f(1.0), name=n))
    for n in _SOFTCORE:
        ET.SubElement(g, 'Parameter', dict(default=_f(values[n]), name=n))
    return e


def _emit_custom_nonbonded(forces, group, energy, per_params, lam_globals, region, nb, particles, exclusions, set1, set2,
                           use_switch, switch_distance, lrc, with_softcore=True):
    periodic = nb.usesPeriodicBoundaryConditions()
    attrs = dict(method=str(2 if periodic else nb.getNonbondedMethod()), cutoff=_f(nb.getCutoffDistance()),
                 useSwitchingFunction=str(int(use_switch)), switchingDistance=_f(switch_distance), useLongRangeCorrection=str(int(lrc)))
    if with_softcore:
        e = _emit_custom(forces, 'CustomNonbondedForce', group, energy, 'PerParticleParameters', per_params, lam_globals, region, **attrs)
    else:
        e = ET.SubElement(forces, 'Force', dict(forceGroup=str(group), name='CustomNonbondedForce', type='CustomNonbondedForce',
                                                energy=energy, version='3', **attrs))
        b = ET.SubElement(e, 'PerParticleParameters')
        for n in per_params:
            ET.SubElement(b, 'Parameter', dict(name=n))
        ET.SubElement(e, 'GlobalParameters')
    ET.SubElement(e, 'ComputedValues'); ET.SubElement(e, 'EnergyParameterDerivatives')
    b = ET.SubElement(e, 'Particles')
    for p in particles:
        ET.SubElement(b, 'Particle', {'param%d' % (k + 1): _f(v) for k, v in enumerate(p)})
    b = ET.SubElement(e, 'Exclusions')
    for (i, j) in exclusions:
        ET.SubElement(b, 'Exclusion', dict(p1=str(i), p2=str(j)))
    ET.SubElement(e, 'Functions')
    groups = ET.SubElement(e, 'InteractionGroups')
    if set1 is not None:
        grp = ET.SubElement(groups, 'InteractionGroup')
        for tag, members in (('Set1', set1), ('Set2', set2)):
            s = ET.SubElement(grp, tag)
            for idx in sorted(members):
                ET.SubElement(s, 'Particle', dict(index=str(idx)))
    return e


def _emit_custom_bond(forces, group, energy, per_params, lam_globals, region, bonds):
    e = _emit_custom(forces, 'CustomBondForce', group, energy, 'PerBondParameters', per_params, lam_globals, region, usesPeriodic='0')
    ET.SubElement(e, 'EnergyParameterDerivatives')
    b = ET.SubElement(e, 'Bonds')
    for (i, j, params) in bonds:
        ET.SubElement(b, 'Bond', dict({'param%d' % (k + 1): _f(v) for k, v in enumerate(params)}, p1=str(i), p2=str(j)))
    return e


def emit_alchemical_forces(forces, system, emit_plain):
    """Append the factory's force set for the marked ``system`` to the <Forces> element.  ``emit_plain(force, **overrides)`` is
    system_xml's emitter of one unmodified force."""
    region = system.alchemical_region
    all_forces = system.getForces()
    nbs = [f for f in all_forces if isinstance(f, NonbondedForce)]
    if len(nbs) != 1:
        raise NotImplementedError('an alchemical System with %d NonbondedForces' % len(nbs))
    nb = nbs[0]
    A = set(region.alchemical_atoms)
    N = set(range(nb.getNumParticles())) - A
    ewald = nb.getNonbondedMethod() in (NonbondedForce.Ewald, NonbondedForce.PME)
    if ewald and not region.annihilate_electrostatics:
        raise NotImplementedError('decoupled electrostatics with the exact PME treatment (alchemy.py:1617-1623 refuses it too)')

    # ---- the NonbondedForce that keeps the non-alchemical interactions ----------------------------------------------
    sig1 = lambda s: 0.1 if s == 0.0 else s                                                  # :1640-1661
    particles = [(q, sig1(s), e) for (q, s, e) in nb.particles]
    exceptions = [(i, j, qq, sig1(s), e) for (i, j, qq, s, e) in nb.exceptions]
    kept_particles = [(0.0, s, 0.0) if k in A else (q, s, e) for k, (q, s, e) in enumerate(particles)]
    kept_exceptions, particle_offsets, exception_offsets = [], [], []
    na_lj, aa_lj, na_qq, aa_qq = [], [], [], []
    for k, (q, s, e) in enumerate(particles):
        if ewald and k in A:
            particle_offsets.append(('lambda_electrostatics', k, q, 0.0, 0.0))
    for n, (i, j, qq, s, e) in enumerate(exceptions):
        n_alch = (i in A) + (j in A)
        if n_alch == 0:
            kept_exceptions.append((i, j, qq, s, e))
            continue
        if ewald and qq != 0.0:
            exception_offsets.append(('lambda_electrostatics', n, qq, 0.0, 0.0))
        if e != 0.0:
            (aa_lj if n_alch == 2 else na_lj).append((i, j, (s, e)))
        if qq != 0.0 and not ewald:
            (aa_qq if n_alch == 2 else na_qq).append((i, j, (qq, s)))
        kept_exceptions.append((i, j, 0.0, s, 0.0))
    exclusions = [(i, j) for (i, j, _, _, _) in exceptions]

    # ---- order and force groups (:1052-1083) ------------------------------------------------------------------------
    untouched = [f for f in all_forces if not isinstance(f, _REMODELLED)]
    readded = [f for f in all_forces if isinstance(f, _REMODELLED) and (not isinstance(f, NonbondedForce) or not ewald)]
    used = {f.getForceGroup() for f in untouched + readded}
    free = sorted(set(range(32)) - used)
    if len(free) < 2:
        raise NotImplementedError('no free force groups for the alchemical forces (alchemy.py:1068-1072 raises here too)')
    g_elec, g_ster = free[0], free[1]

    rf_replaced = nb.getNonbondedMethod() == NonbondedForce.CutoffPeriodic and getattr(system, 'rf_unshifted_switch_width', None) is not None
    if rf_replaced:                                                          # forcefactories.py:82-84: charges move to the last force
        kept_particles = [(0.0, s, e) for (q, s, e) in kept_particles]

    def emit_nb(group):
        emit_plain(nb, particles=kept_particles, exceptions=kept_exceptions, force_group=group,
                   global_parameters=[('lambda_electrostatics', 1.0)] if ewald else [],
                   particle_offsets=particle_offsets, exception_offsets=exception_offsets)

    for f in untouched:
        emit_plain(f)
    for f in readded:
        if isinstance(f, NonbondedForce):
            emit_nb(f.getForceGroup())
        else:
            emit_plain(f)

    lrc = nb.getUseDispersionCorrection() and getattr(system, 'alchemical_lrc', True)
    cnb = dict(region=region, nb=nb, exclusions=exclusions)
    if not ewald:                                                            # 'lambda_electrostatics' sorts first
        pair_expr, exc_expr = electrostatics_expressions(nb)
        fixed = '' if region.annihilate_electrostatics else 'lambda_electrostatics=1.0;'
        charges = [(q, s) for (q, s, e) in particles]
        switched = nb.getNonbondedMethod() in (NonbondedForce.CutoffPeriodic, NonbondedForce.CutoffNonPeriodic)
        switch_width = getattr(system, 'rf_unshifted_switch_width', None) or 0.1
        common = dict(use_switch=switched, switch_distance=nb.getCutoffDistance() - switch_width, lrc=False, **cnb)
        _emit_custom_nonbonded(forces, g_elec, pair_expr, ('charge', 'sigma'), ('lambda_electrostatics',), particles=charges,
                               set1=N, set2=A, **common)
        _emit_custom_nonbonded(forces, g_elec, pair_expr + fixed, ('charge', 'sigma'), () if fixed else ('lambda_electrostatics',),
                               particles=charges, set1=A, set2=A, **common)
        _emit_custom_bond(forces, g_elec, exc_expr, ('chargeprod', 'sigma'), ('lambda_electrostatics',), region, na_qq)
        _emit_custom_bond(forces, g_elec, exc_expr + fixed, ('chargeprod', 'sigma'), () if fixed else ('lambda_electrostatics',), region, aa_qq)
    else:
        emit_nb(g_elec)
    exc_expr = sterics_exception_expression()
    pair_expr = exc_expr + _MIX_STERICS
    fixed = '' if region.annihilate_sterics else 'lambda_sterics=1.0;'
    lj = [(s, e) for (q, s, e) in particles]
    common = dict(use_switch=nb.getUseSwitchingFunction(), switch_distance=nb.getSwitchingDistance(), lrc=lrc, **cnb)
    _emit_custom_nonbonded(forces, g_ster, pair_expr, ('sigma', 'epsilon'), ('lambda_sterics',), particles=lj, set1=N, set2=A, **common)
    _emit_custom_nonbonded(forces, g_ster, pair_expr + fixed, ('sigma', 'epsilon'), () if fixed else ('lambda_sterics',),
                           particles=lj, set1=A, set2=A, **common)
    _emit_custom_bond(forces, g_ster, exc_expr, ('sigma', 'epsilon'), ('lambda_sterics',), region, na_lj)
    _emit_custom_bond(forces, g_ster, exc_expr + fixed, ('sigma', 'epsilon'), () if fixed else ('lambda_sterics',), region, aa_lj)
    if rf_replaced:                                                          # forcefactories.py:76-84: the environment's charges live here
        eps, rc = nb.getReactionFieldDielectric(), nb.getCutoffDistance()
        energy = _RF_HEAD + 'chargeprod = charge1*charge2;k_rf = %f;ONE_4PI_EPS0 = %f;' % (rc ** -3 * (eps - 1.0) / (2.0 * eps + 1.0), ONE_4PI_EPS0)
        switch_width = getattr(system, 'rf_unshifted_switch_width', None) or 0.1
        _emit_custom_nonbonded(forces, 0, energy, ('charge',), (), region, nb, [(q,) for (q, s, e) in particles], exclusions, None, None,
                               True, rc - switch_width, False, with_softcore=False)


# softened bonded terms: class -> (energy expression with the lambda's name, per-term parameters), alchemy.py:1341, 1261, 1180
_BONDED_CUSTOM = {'bond': ('%s*(K/2)*(r-r0)^2;', ('r0', 'K')),
                  'angle': ('%s*(K/2)*(theta-theta0)^2;', ('theta0', 'K')),
                  'torsion': ('%s*k*(1+cos(periodicity*theta-phase))', ('periodicity', 'phase', 'k'))}
_BONDED_TAGS = {'bond': ('CustomBondForce', 'PerBondParameters', 'Bonds', 'Bond'), 'angle': ('CustomAngleForce', 'PerAngleParameters', 'Angles', 'Angle'),
                'torsion': ('CustomTorsionForce', 'PerTorsionParameters', 'Torsions', 'Torsion')}


def _emit_custom_bonded(forces, group, kind, energy, per, lam, terms):
    ftype, per_tag, list_tag, item_tag = _BONDED_TAGS[kind]
    e = ET.SubElement(forces, 'Force', dict(forceGroup=str(group), name=ftype, type=ftype, energy=energy, version='3', usesPeriodic='0'))
    b = ET.SubElement(e, per_tag)
    for n in per:
        ET.SubElement(b, 'Parameter', dict(name=n))
    g = ET.SubElement(e, 'GlobalParameters')
    ET.SubElement(g, 'Parameter', dict(default=_f(1.0), name=lam))
    ET.SubElement(e, 'EnergyParameterDerivatives')
    b = ET.SubElement(e, list_tag)
    for atoms, params in terms:
        ET.SubElement(b, item_tag, dict({'param%d' % (k + 1): _f(v) for k, v in enumerate(params)}, **{'p%d' % (k + 1): str(a) for k, a in enumerate(atoms)}))
    return e


# ---- alchemical GBSA: the CustomGBForce the factory builds from a GBSAOBCForce (alchemy.py:2172-2225), its strings verbatim -----------------
_GB_I = ("(lambda_electrostatics*alchemical2 + (1-alchemical2))*step(r+sr2-or1)*0.5*(1/L-1/U+0.25*(r-sr2^2/r)*(1/(U^2)-1/(L^2))+0.5*log(L/U)/r+C);"
         "U=r+sr2;"
         "C=2*(1/or1-1/L)*step(sr2-r-or1);"
         "L=max(or1, D);"
         "D=abs(r-sr2);"
         "sr2 = scale2*or2;"
         "or1 = radius1-offset; or2 = radius2-offset")
_GB_B = ("1/(1/or-tanh(psi-0.8*psi^2+4.85*psi^3)/radius);"
         "psi=I*or; or=radius-offset")
_GB_SELF = "-0.5*138.935485*(1/soluteDielectric-1/solventDielectric)*(lambda_electrostatics*alchemical+(1-alchemical))*charge^2/B"
_GB_SURFACE = "(lambda_electrostatics*alchemical+(1-alchemical))*28.3919551*(radius+0.14)^2*(radius/B)^6"
_GB_PAIR = ("-138.935485*(1/soluteDielectric-1/solventDielectric)*(lambda_electrostatics*alchemical1+(1-alchemical1))*charge1*(lambda_electrostatics*alchemical2+(1-alchemical2))*charge2/f;"
            "f=sqrt(r^2+B1*B2*exp(-r^2/(4*B1*B2)))")
# OpenMM's CustomGBForce::ComputationType
_GB_SINGLE, _GB_PAIR_NOEXCL = 0, 2


def _emit_custom_gb(forces, group, gb, alchemical_atoms):
    """CustomGBForce element (OpenMM's CustomGBForceProxy layout: external knowledge, unpinned like the other custom forces here)"""
    e = ET.SubElement(forces, 'Force', dict(forceGroup=str(group), name='CustomGBForce', type='CustomGBForce', method=str(gb.getNonbondedMethod()),
                                            cutoff=_f(1.0), version='2'))
    b = ET.SubElement(e, 'PerParticleParameters')
    for n in ('charge', 'radius', 'scale', 'alchemical'):
        ET.SubElement(b, 'Parameter', dict(name=n))
    g = ET.SubElement(e, 'GlobalParameters')
    for n, v in (('lambda_electrostatics', 1.0), ('solventDielectric', gb.getSolventDielectric()), ('soluteDielectric', gb.getSoluteDielectric()), ('offset', 0.009)):
        ET.SubElement(g, 'Parameter', dict(default=_f(v), name=n))
    ET.SubElement(e, 'EnergyParameterDerivatives')
    b = ET.SubElement(e, 'Particles')
    A = set(alchemical_atoms)
    for k, (q, r, sc) in enumerate(gb.particles):
        ET.SubElement(b, 'Particle', dict(param1=_f(q), param2=_f(r), param3=_f(sc), param4=_f(1.0 if k in A else 0.0)))
    ET.SubElement(e, 'Exclusions'); ET.SubElement(e, 'Functions')
    b = ET.SubElement(e, 'ComputedValues')
    ET.SubElement(b, 'Value', dict(name='I', expression=_GB_I, type=str(_GB_PAIR_NOEXCL)))
    ET.SubElement(b, 'Value', dict(name='B', expression=_GB_B, type=str(_GB_SINGLE)))
    b = ET.SubElement(e, 'EnergyTerms')
    ET.SubElement(b, 'Term', dict(expression=_GB_SELF, type=str(_GB_SINGLE)))
    if gb.getSurfaceAreaEnergy() != 0.0:
        ET.SubElement(b, 'Term', dict(expression=_GB_SURFACE, type=str(_GB_SINGLE)))
    ET.SubElement(b, 'Term', dict(expression=_GB_PAIR, type=str(_GB_PAIR_NOEXCL)))
    return e


def parse_custom_gb(e):
    return dict(type='CustomGBForce', energy='', group=int(e.get('forceGroup', '0')), attrs=dict(e.attrib),
                globals={g.get('name'): float(g.get('default')) for g in _kids(e, 'GlobalParameters')},
                per=[p.get('name') for p in _kids(e, 'PerParticleParameters')], particles=[_params(p) for p in _kids(e, 'Particles')],
                values=[(v.get('name'), v.get('expression'), int(v.get('type'))) for v in _kids(e, 'ComputedValues')],
                terms=[(t.get('expression'), int(t.get('type'))) for t in _kids(e, 'EnergyTerms')])


def gbsa_from_custom_gb(c):
    """the GBSAOBCForce a parsed alchemical CustomGBForce came from + its alchemical atoms (only the factory's own expressions are understood)"""
    from .system import GBSAOBCForce
    compact = lambda s: s.replace(' ', '')
    if c['per'] != ['charge', 'radius', 'scale', 'alchemical'] or [compact(v[1]) for v in c['values']] != [compact(_GB_I), compact(_GB_B)]:
        raise NotImplementedError('CustomGBForce other than the alchemical factory\'s GBSA (OBC2): %s' % (c['values'][0][1][:60] if c['values'] else '-'))
    terms = [compact(t[0]) for t in c['terms']]
    if terms not in ([compact(_GB_SELF), compact(_GB_SURFACE), compact(_GB_PAIR)], [compact(_GB_SELF), compact(_GB_PAIR)]):
        raise NotImplementedError('CustomGBForce energy terms other than the alchemical factory\'s GBSA')
    gb = GBSAOBCForce()
    gb.setNonbondedMethod(int(c['attrs'].get('method', '0')))
    gb.setSolventDielectric(c['globals'].get('solventDielectric', 78.5)); gb.setSoluteDielectric(c['globals'].get('soluteDielectric', 1.0))
    gb.setSurfaceAreaEnergy(2.25936 if len(terms) == 3 else 0.0)
    atoms = []
    for k, p in enumerate(c['particles']):
        gb.addParticle(p[0], p[1], p[2])
        if p[3] != 0.0:
            atoms.append(k)
    return gb, atoms


def emit_region_forces(forces, system, emit_plain):
    """The force set of a System in the general-regions mode (``system.alchemical_regions``): the reference's loop over
    single_regions + pair_regions (alchemy.py:1693-2036) restated on plain tables -- INCLUDING its order of reading and zeroing the
    NonbondedForce's parameters, which decides what the later forces' particle tables hold (:1886-1911, 2001-2006)."""
    from .system import GBSAOBCForce
    gbs = [f for f in system.getForces() if isinstance(f, GBSAOBCForce)]
    regions = system.alchemical_regions
    opts = system.alchemical_factory_options
    terms = system.alchemical_region_terms
    all_forces = system.getForces()
    nbs = [f for f in all_forces if isinstance(f, NonbondedForce)]
    if len(nbs) != 1:
        raise NotImplementedError('an alchemical System with %d NonbondedForces' % len(nbs))
    nb = nbs[0]
    ewald = nb.getNonbondedMethod() in (NonbondedForce.Ewald, NonbondedForce.PME)
    exact = ewald and opts['alchemical_pme_treatment'] == 'exact'
    n = nb.getNumParticles()
    # the reference NonbondedForce again (sigma = 0 already 0.1 nm): the kept force + what the custom forces took
    cur = [(float(terms['charge'][k]), float(terms['sigma'][k]), float(terms['epsilon'][k])) for k in range(n)]
    took = {frozenset((int(i), int(j))): tuple(float(v) for v in p) for (i, j), p in zip(terms['exception_atoms'], terms['exception_params'])}
    exc = [(i, j) + took.get(frozenset((i, j)), (qq, sg, ep)) for (i, j, qq, sg, ep) in nb.exceptions]
    if exact:                                                                # regions that do not interact exclude each other (:1663-1672)
        together = set(getattr(system, 'alchemical_regions_interactions', []))
        have = {frozenset((i, j)): k for k, (i, j, _, _, _) in enumerate(exc)}
        for x in range(len(regions)):
            for y in range(x + 1, len(regions)):
                if (x, y) in together:
                    continue
                for a1 in regions[x].alchemical_atoms:
                    for a2 in regions[y].alchemical_atoms:
                        k = have.get(frozenset((a1, a2)))
                        if k is None:
                            exc.append((a1, a2, 0.0, 1.0, 0.0))
                        else:
                            exc[k] = (exc[k][0], exc[k][1], 0.0, 1.0, 0.0)
    exclusions = [(i, j) for (i, j, _, _, _) in exc]
    alch_all = set()
    for r in regions:
        alch_all |= set(r.alchemical_atoms)
    env = set(range(n)) - alch_all
    suffix = lambda r: '' if r.name is None else '_' + r.name
    by_lambda, particle_offsets, exception_offsets, nb_globals = {}, [], [], []
    if exact:
        nb_globals = [('lambda_electrostatics' + suffix(r), 1.0) for r in regions]
    singles = [[r] for r in regions]
    pairs = [[regions[a], regions[b]] for a, b in getattr(system, 'alchemical_regions_interactions', [])]
    lrc = nb.getUseDispersionCorrection() and getattr(system, 'alchemical_lrc', True)
    switched_e = (ewald and opts['alchemical_pme_treatment'] == 'coulomb') or \
        (not ewald and opts['alchemical_rf_treatment'] == 'switched' and nb.getNonbondedMethod() != NonbondedForce.NoCutoff)
    fixed = lambda names, on: '' if on else ''.join(x + '=1.0;' for x in names)
    last_key = None
    for group_regions in singles + pairs:
        sfx = [suffix(r) for r in group_regions]
        region = group_regions[-1]                                           # whose softcore_* the forces get (:2009-2025)
        lam_s = 'lambda_sterics' + sfx[0] if len(sfx) == 1 else 'lambda_sterics%s*lambda_sterics%s' % (sfx[0], sfx[1])
        lam_e = 'lambda_electrostatics' + sfx[0] if len(sfx) == 1 else 'lambda_electrostatics%s*lambda_electrostatics%s' % (sfx[0], sfx[1])
        names_s, names_e = ['lambda_sterics' + x for x in sfx], ['lambda_electrostatics' + x for x in sfx]
        A0, A1 = set(group_regions[0].alchemical_atoms), set(group_regions[-1].alchemical_atoms)
        lj = [(sg, ep) for (q, sg, ep) in cur]                               # read BEFORE this turn zeroes its region (:1886-1899)
        charges = [(q, sg) for (q, sg, ep) in cur]
        if exact and len(sfx) == 1:
            particle_offsets += [(names_e[0], k, cur[k][0], 0.0, 0.0) for k in sorted(A0)]
        for k in A0:
            cur[k] = (0.0, cur[k][1], 0.0)                                   # :1903-1911
        na_lj, aa_lj, na_qq, aa_qq = [], [], [], []
        for idx, (i, j, qq, sg, ep) in enumerate(exc):
            if len(sfx) > 1:
                both = (i in A0 and j in A1) or (j in A0 and i in A1)
                one = any_ = False
            else:
                both = i in A0 and j in A0
                any_ = i in A0 or j in A0
                one = any_ and not both
                if exact and any_ and qq != 0.0:
                    exception_offsets.append((names_e[0], idx, qq, 0.0, 0.0))
            if both:
                if ep != 0.0:
                    aa_lj.append((i, j, (sg, ep)))
                if qq != 0.0 and not exact:
                    aa_qq.append((i, j, (qq, sg)))
            elif one:
                if ep != 0.0:
                    na_lj.append((i, j, (sg, ep)))
                if qq != 0.0 and not exact:
                    na_qq.append((i, j, (qq, sg)))
            if any_:
                exc[idx] = (i, j, 0.0, sg, 0.0)                              # :2001-2006
        ster, elec = [], []
        exc_expr = sterics_exception_expression(lam_s)
        pair_expr = exc_expr + _MIX_STERICS
        common = dict(use_switch=nb.getUseSwitchingFunction(), switch_distance=nb.getSwitchingDistance(), lrc=lrc, region=region, nb=nb,
                      exclusions=exclusions, particles=lj)
        if len(sfx) > 1:
            ster.append(('nb', dict(energy=pair_expr, per_params=('sigma', 'epsilon'), lam_globals=tuple(names_s), set1=A0, set2=A1, **common)))
            ster.append(('bond', dict(energy=exc_expr, per_params=('sigma', 'epsilon'), lam_globals=tuple(names_s), region=region, bonds=aa_lj)))
        else:
            fx = fixed(names_s, region.annihilate_sterics)
            ster.append(('nb', dict(energy=pair_expr, per_params=('sigma', 'epsilon'), lam_globals=tuple(names_s), set1=env, set2=A0, **common)))
            ster.append(('nb', dict(energy=pair_expr + fx, per_params=('sigma', 'epsilon'), lam_globals=() if fx else tuple(names_s), set1=A0, set2=A0, **common)))
            ster.append(('bond', dict(energy=exc_expr, per_params=('sigma', 'epsilon'), lam_globals=tuple(names_s), region=region, bonds=na_lj)))
            ster.append(('bond', dict(energy=exc_expr + fx, per_params=('sigma', 'epsilon'), lam_globals=() if fx else tuple(names_s), region=region, bonds=aa_lj)))
        if not exact:
            e_pair, e_exc = electrostatics_expressions(nb, lam_e, opts['alchemical_pme_treatment'], opts['alchemical_rf_treatment'], opts.get('consistent_exceptions', False))
            common = dict(use_switch=switched_e, switch_distance=nb.getCutoffDistance() - opts['switch_width'], lrc=False, region=region, nb=nb,
                          exclusions=exclusions, particles=charges)
            if len(sfx) > 1:
                elec.append(('nb', dict(energy=e_pair, per_params=('charge', 'sigma'), lam_globals=tuple(names_e), set1=A0, set2=A1, **common)))
                elec.append(('bond', dict(energy=e_exc, per_params=('chargeprod', 'sigma'), lam_globals=tuple(names_e), region=region, bonds=aa_qq)))
            else:
                fx = fixed(names_e, region.annihilate_electrostatics)
                elec.append(('nb', dict(energy=e_pair, per_params=('charge', 'sigma'), lam_globals=tuple(names_e), set1=env, set2=A0, **common)))
                elec.append(('nb', dict(energy=e_pair + fx, per_params=('charge', 'sigma'), lam_globals=() if fx else tuple(names_e), set1=A0, set2=A0, **common)))
                elec.append(('bond', dict(energy=e_exc, per_params=('chargeprod', 'sigma'), lam_globals=tuple(names_e), region=region, bonds=na_qq)))
                elec.append(('bond', dict(energy=e_exc + fx, per_params=('chargeprod', 'sigma'), lam_globals=() if fx else tuple(names_e), region=region, bonds=aa_qq)))
        by_lambda.setdefault('lambda_electrostatics' + sfx[0], []).extend(elec)              # :2027-2032
        by_lambda.setdefault('lambda_sterics' + sfx[0], []).extend(ster)
        last_key = 'lambda_electrostatics' + sfx[0]
    # softened bonded terms: one Custom{Bond,Angle,Torsion}Force per region and class, energy lambda x the reference term (:1170-1197, 1252-1275, 1331-1354)
    for kind, (expr, per) in _BONDED_CUSTOM.items():
        atoms = terms.get(kind + '_atoms')
        if atoms is None or len(atoms) == 0:
            continue
        for g, r in enumerate(regions):
            sel = [k for k in range(len(atoms)) if int(terms[kind + '_region'][k]) == g + 1]
            if sel:
                name = 'lambda_%ss%s' % (kind, suffix(r))
                by_lambda.setdefault(name, []).append(('bonded', dict(kind=kind, energy=expr % name, per=per, lam=name,
                                                                      terms=[(tuple(int(a) for a in atoms[k]), tuple(float(v) for v in terms[kind + '_params'][k])) for k in sel])))
    # ---- order and force groups (:1052-1083) ------------------------------------------------------------------------
    for gb in gbs:                                                           # the factory's CustomGBForce joins the lambda_electrostatics forces (:2225)
        by_lambda.setdefault('lambda_electrostatics' + suffix(regions[0]), []).append(('gb', dict(gb=gb, alchemical_atoms=regions[0].alchemical_atoms)))
    untouched = [f for f in all_forces if not isinstance(f, _REMODELLED) and not isinstance(f, GBSAOBCForce)]
    readded = [f for f in all_forces if isinstance(f, _REMODELLED) and (not isinstance(f, NonbondedForce) or not exact)]
    free = sorted(set(range(32)) - {f.getForceGroup() for f in untouched + readded})
    if len(free) < len(by_lambda):
        raise NotImplementedError('no free force groups for the alchemical forces (alchemy.py:1068-1072 raises here too)')
    rf_replaced = nb.getNonbondedMethod() == NonbondedForce.CutoffPeriodic and opts['alchemical_rf_treatment'] == 'switched'
    kept_particles = [(0.0, sg, ep) for (q, sg, ep) in cur] if rf_replaced else list(cur)   # forcefactories.py:82-84: charges move to the last force

    def emit_nb(group):
        emit_plain(nb, particles=kept_particles, exceptions=exc, force_group=group, global_parameters=nb_globals,
                   particle_offsets=particle_offsets, exception_offsets=exception_offsets)
    for f in untouched:
        emit_plain(f)
    for f in readded:
        if isinstance(f, NonbondedForce):
            emit_nb(f.getForceGroup())
        else:
            emit_plain(f)
    for key in sorted(by_lambda):
        group = free.pop(0)
        for kind, kw in by_lambda[key]:
            if kind == 'nb':
                _emit_custom_nonbonded(forces, group, kw.pop('energy'), kw.pop('per_params'), kw.pop('lam_globals'), **kw)
            elif kind == 'bonded':
                _emit_custom_bonded(forces, group, **kw)
            elif kind == 'gb':
                _emit_custom_gb(forces, group, **kw)
            else:
                _emit_custom_bond(forces, group, kw['energy'], kw['per_params'], kw['lam_globals'], kw['region'], kw['bonds'])
        if exact and key == last_key:
            emit_nb(group)                                                   # :2034-2035
    if rf_replaced:
        eps, rc = nb.getReactionFieldDielectric(), nb.getCutoffDistance()
        energy = _RF_HEAD + 'chargeprod = charge1*charge2;k_rf = %f;ONE_4PI_EPS0 = %f;' % (rc ** -3 * (eps - 1.0) / (2.0 * eps + 1.0), ONE_4PI_EPS0)
        _emit_custom_nonbonded(forces, 0, energy, ('charge',), (), regions[-1], nb, [(q,) for (q, sg, ep) in cur], exclusions, None, None,
                               True, rc - opts['switch_width'], False, with_softcore=False)


# ---- reader -------------------------------------------------------------------------------------------------------
def _params(elem):
    out, k = [], 1
    while elem.get('param%d' % k) is not None:
        out.append(float(elem.get('param%d' % k)))
        k += 1
    return out


def _kids(e, tag):
    b = e.find(tag)
    return [] if b is None else list(b)


def parse_custom(e):
    """A CustomNonbondedForce / CustomBondForce element as a plain dictionary."""
    d = dict(type=e.get('type'), energy=e.get('energy', ''), group=int(e.get('forceGroup', '0')), attrs=dict(e.attrib),
             globals={g.get('name'): float(g.get('default')) for g in _kids(e, 'GlobalParameters')})
    if d['type'] == 'CustomNonbondedForce':
        d['per'] = [p.get('name') for p in _kids(e, 'PerParticleParameters')]
        d['particles'] = [_params(p) for p in _kids(e, 'Particles')]
        d['groups'] = [tuple(sorted(int(p.get('index')) for p in _kids(g, tag)) for tag in ('Set1', 'Set2'))
                       for g in _kids(e, 'InteractionGroups')]
    elif d['type'] == 'CustomBondForce':
        d['per'] = [p.get('name') for p in _kids(e, 'PerBondParameters')]
        d['bonds'] = [(int(b.get('p1')), int(b.get('p2')), _params(b)) for b in _kids(e, 'Bonds')]
    else:                                                    # CustomAngleForce / CustomTorsionForce: softened bonded terms only
        width, per_tag, list_tag = (3, 'PerAngleParameters', 'Angles') if d['type'] == 'CustomAngleForce' else (4, 'PerTorsionParameters', 'Torsions')
        d['per'] = [p.get('name') for p in _kids(e, per_tag)]
        d['terms'] = [(tuple(int(b.get('p%d' % (k + 1))) for k in range(width)), _params(b)) for b in _kids(e, list_tag)]
    return d


def is_alchemical_document(nb_offsets, customs):
    return bool(nb_offsets) or any('softcore_alpha' in c['globals'] for c in customs)


def rebuild_marked_system(system, nb, global_parameters, particle_offsets, exception_offsets, customs):
    """Undo the factory on a parsed document: give the NonbondedForce back the alchemical atoms' charges, Lennard-Jones
    parameters and exceptions, drop the custom forces, and mark the region on ``system``.  ``nb`` is the parsed (kept)
    NonbondedForce, the offsets are (parameter, index, q, sig, eps) tuples, ``customs`` the parse_custom dictionaries."""
    from .alchemy import AlchemicalRegion
    compact = lambda s: s.replace(' ', '')
    sterics = [c for c in customs if 'U_sterics' in c['energy']]
    electro = [c for c in customs if 'U_electrostatics' in c['energy']]
    rf = [c for c in customs if compact(c['energy']).startswith(compact(_RF_HEAD)) and c['type'] == 'CustomNonbondedForce']
    bonded = [c for c in customs if re.match(r'lambda_(bonds|angles|torsions)\w*\*', compact(c['energy']))]
    other = [c for c in customs if c not in sterics + electro + rf + bonded and c['type'] != 'CustomGBForce']
    if other:
        raise NotImplementedError('custom force outside the alchemical factory\'s set: %s' % other[0]['energy'][:60])
    for name in list(global_parameters) + [p[0] for p in particle_offsets + exception_offsets]:
        if not name.startswith('lambda_electrostatics'):
            raise NotImplementedError('NonbondedForce offset parameter %r (only the alchemical factory\'s lambda_electrostatics[_<region>])' % name)
    if needs_general_reader(nb, global_parameters, particle_offsets, exception_offsets, customs):
        # several / named regions, soft-core electrostatics, the 'direct-space' / 'coulomb' PME treatments, a charged region under a
        # reaction-field method: back to the reference System and through this package's factory
        return rebuild_general_system(system, nb, global_parameters, particle_offsets, exception_offsets, customs)
    for name in list(global_parameters) + [p[0] for p in particle_offsets + exception_offsets]:
        if name != 'lambda_electrostatics':
            raise NotImplementedError('NonbondedForce offset parameter %r (one unnamed alchemical region only)' % name)
    for c in sterics + electro:
        for g in c['globals']:
            if g not in _SOFTCORE + ('lambda_sterics', 'lambda_electrostatics'):
                raise NotImplementedError('alchemical parameter %r (one unnamed region, no bonded lambdas)' % g)
    cnb_s = [c for c in sterics if c['type'] == 'CustomNonbondedForce']
    na = [c for c in cnb_s if 'lambda_sterics' in c['globals'] and c['groups'] and c['groups'][0][0] != c['groups'][0][1]]
    aa = [c for c in cnb_s if c['groups'] and c['groups'][0][0] == c['groups'][0][1]]
    if len(na) != 1 or len(aa) != 1 or len(cnb_s) != 2:
        raise NotImplementedError('sterics CustomNonbondedForces of more than one alchemical region')
    na, aa = na[0], aa[0]
    A = list(aa['groups'][0][0])
    if sorted(na['groups'][0][1]) != sorted(A):
        raise NotImplementedError('interaction groups of the two sterics forces disagree on the alchemical atoms')
    annihilate_sterics = 'lambda_sterics' in aa['globals']
    if compact(sterics_exception_expression() + _MIX_STERICS) != compact(na['energy']):
        raise NotImplementedError('sterics expression other than the factory\'s soft-core Lennard-Jones: %s' % na['energy'][:80])
    g = na['globals']
    if (g.get('softcore_beta', 0.0), g.get('softcore_d', 1.0), g.get('softcore_e', 1.0), g.get('softcore_f', 2.0)) != (0.0, 1.0, 1.0, 2.0):
        raise NotImplementedError('soft-core electrostatics')
    cnb_e = [c for c in electro if c['type'] == 'CustomNonbondedForce']
    annihilate_electrostatics = True
    for c in cnb_e:
        if c['groups'] and c['groups'][0][0] == c['groups'][0][1] and 'lambda_electrostatics' not in c['globals']:
            annihilate_electrostatics = False
    region = AlchemicalRegion(alchemical_atoms=A, annihilate_electrostatics=annihilate_electrostatics, annihilate_sterics=annihilate_sterics,
                              softcore_alpha=g['softcore_alpha'], softcore_a=g['softcore_a'], softcore_b=g['softcore_b'], softcore_c=g['softcore_c'])
    Aset = set(A)
    # particles
    charge = {p[1]: p[2] for p in particle_offsets}
    if cnb_e:
        for k in A:
            charge[k] = cnb_e[0]['particles'][k][0]
    for k in A:
        sig, eps = na['particles'][k]
        nb.setParticleParameters(k, charge.get(k, 0.0), sig, eps)
    if rf:                                                                   # the factory moved ALL charges there
        for k, p in enumerate(rf[0]['particles']):
            if k not in Aset:
                q, s, e = nb.getParticleParameters(k)
                nb.setParticleParameters(k, p[0], s, e)
    # exceptions
    qq_of = {p[1]: p[2] for p in exception_offsets}
    lj_of, qq_bond = {}, {}
    for c in sterics:
        if c['type'] == 'CustomBondForce':
            for (i, j, p) in c['bonds']:
                lj_of[frozenset((i, j))] = (p[0], p[1])
    for c in electro:
        if c['type'] == 'CustomBondForce':
            for (i, j, p) in c['bonds']:
                qq_bond[frozenset((i, j))] = p[0]
    for n, (i, j, qq, s, e) in enumerate(list(nb.exceptions)):
        if i in Aset or j in Aset:
            key = frozenset((i, j))
            sig, eps = lj_of.get(key, (s, 0.0))
            nb.exceptions[n] = (i, j, qq_of.get(n, qq_bond.get(key, 0.0)), sig, eps)
    system.alchemical_region = region
    if rf:
        system.rf_unshifted_switch_width = round(nb.getCutoffDistance() - float(rf[0]['attrs'].get('switchingDistance', nb.getCutoffDistance() - 0.1)), 12)
    use_lrc = na['attrs'].get('useLongRangeCorrection', '0') not in ('0', 'false')
    system.alchemical_lrc = use_lrc or not nb.getUseDispersionCorrection()
    if particle_offsets or global_parameters:                                # the factory put it into the lambda_electrostatics group
        nb.setForceGroup(0)
    return system


# ---- reader of the general force set (several / named regions, soft-core electrostatics, the non-exact PME treatments) -------------
_LAMBDA_OF = re.compile(r'\(\((.*?)\)\^softcore_[ad]\)')


def _lambda_names(c):
    """the lambda variables of a custom force's expression, e.g. ['lambda_sterics_zero', 'lambda_sterics_one']"""
    m = _LAMBDA_OF.search(c['energy'].replace(' ', ''))
    if m is None:
        raise NotImplementedError('custom force without a lambda^softcore prefactor: %s' % c['energy'][:60])
    return m.group(1).split('*')


def needs_general_reader(nb, global_parameters, particle_offsets, exception_offsets, customs):
    """True when the document holds more than the one unnamed region under the exact PME treatment / without alchemical charges that
    rebuild_marked_system undoes itself"""
    if any(c['type'] == 'CustomGBForce' for c in customs) or nb.getNonbondedMethod() == NonbondedForce.NoCutoff:
        return True
    if any(re.match(r'lambda_(bonds|angles|torsions)', c['energy'].replace(' ', '')) for c in customs):
        return True
    names = set(global_parameters) | {p[0] for p in particle_offsets + exception_offsets}
    sterics = [c for c in customs if 'U_sterics' in c['energy']]
    electro = [c for c in customs if 'U_electrostatics' in c['energy']]
    for c in sterics + electro:
        names |= set(_lambda_names(c))
        g = c['globals']
        if (g.get('softcore_beta', 0.0), g.get('softcore_d', 1.0), g.get('softcore_e', 1.0), g.get('softcore_c', 6.0)) != (0.0, 1.0, 1.0, 6.0):
            return True
    if any(n not in ('lambda_sterics', 'lambda_electrostatics') for n in names):
        return True
    if len([c for c in sterics if c['type'] == 'CustomNonbondedForce']) != 2:
        return True
    ewald = nb.getNonbondedMethod() in (NonbondedForce.Ewald, NonbondedForce.PME)
    charged = any(p[0] != 0.0 for c in electro if c['type'] == 'CustomNonbondedForce' for p in c['particles'])
    return bool(electro) and (ewald or charged)


def rebuild_general_system(system, nb, global_parameters, particle_offsets, exception_offsets, customs):
    """Undo the factory on a parsed document of the general force set -- back to (reference NonbondedForce, regions, interacting pairs,
    factory options) -- and run this package's factory on that (alchemy.AbsoluteAlchemicalFactory.create_alchemical_system): the marked
    System it returns replaces ``system``."""
    from .alchemy import AlchemicalRegion, AbsoluteAlchemicalFactory
    compact = lambda s: s.replace(' ', '')
    for c in [c for c in customs if c['type'] == 'CustomGBForce']:            # the alchemical GBSA: back to the GBSAOBCForce it came from
        gb, gb_atoms = gbsa_from_custom_gb(c)
        system.addForce(gb)
    customs = [c for c in customs if c['type'] != 'CustomGBForce']
    sterics = [c for c in customs if 'U_sterics' in c['energy']]
    electro = [c for c in customs if 'U_electrostatics' in c['energy']]
    rf = [c for c in customs if compact(c['energy']).startswith(compact(_RF_HEAD)) and c['type'] == 'CustomNonbondedForce']
    bonded = [c for c in customs if re.match(r'lambda_(bonds|angles|torsions)\w*\*', compact(c['energy']))]
    other = [c for c in customs if c not in sterics + electro + rf + bonded]
    if other:
        raise NotImplementedError('custom force outside the alchemical factory\'s set: %s' % other[0]['energy'][:60])
    sfx_of = lambda name: name[len('lambda_sterics'):] if name.startswith('lambda_sterics') else name[len('lambda_electrostatics'):]
    for c in sterics + electro:
        c['lam'] = _lambda_names(c)
        c['sfx'] = [sfx_of(x) for x in c['lam']]
        for g in c['globals']:
            if g not in _SOFTCORE and not g.startswith(('lambda_sterics', 'lambda_electrostatics')):
                raise NotImplementedError('alchemical parameter %r (no bonded lambdas)' % g)
    is_nb = lambda c: c['type'] == 'CustomNonbondedForce'
    single_s = [c for c in sterics if is_nb(c) and len(c['sfx']) == 1]
    na_s = {c['sfx'][0]: c for c in single_s if c['groups'] and c['groups'][0][0] != c['groups'][0][1]}
    aa_s = {c['sfx'][0]: c for c in single_s if c['groups'] and c['groups'][0][0] == c['groups'][0][1]}
    if not na_s or set(na_s) != set(aa_s) or len(single_s) != 2 * len(na_s):
        raise NotImplementedError('sterics CustomNonbondedForces that are not one (environment, region) + one (region, region) force per region')
    elec_sfx = {sfx_of(x) for x in list(global_parameters) + [p[0] for p in particle_offsets + exception_offsets]} | {x for c in electro for x in c['sfx']}
    if not elec_sfx <= set(na_s):
        raise NotImplementedError('lambda_electrostatics%s of a region without sterics forces (regions: %s)' % (sorted(elec_sfx - set(na_s))[0], sorted(na_s)))
    single_e = [c for c in electro if is_nb(c) and len(c['sfx']) == 1]
    na_e = {c['sfx'][0]: c for c in single_e if c['groups'] and c['groups'][0][0] != c['groups'][0][1]}
    aa_e = {c['sfx'][0]: c for c in single_e if c['groups'] and c['groups'][0][0] == c['groups'][0][1]}
    atoms = {x: sorted(na_s[x]['groups'][0][1]) for x in na_s}
    owner = {k: x for x, A in atoms.items() for k in A}
    # bonds by the region whose single forces hold them
    bonds_s, bonds_e = {}, {}
    for c in sterics + electro:
        if not is_nb(c) and len(c['sfx']) == 1:
            for (i, j, p) in c['bonds']:
                (bonds_s if c in sterics else bonds_e).setdefault(c['sfx'][0], []).append((i, j, p))
    # the factory's order of the regions: by name, except that an exception between two regions sits in the EARLIER region's forces
    order = sorted(na_s)
    before = set()
    for x, bl in list(bonds_s.items()) + list(bonds_e.items()):
        for (i, j, p) in bl:
            for k in (i, j):
                if k in owner and owner[k] != x:
                    before.add((x, owner[k]))
    for _ in range(len(order) ** 2):
        bad = next(((a, b) for (a, b) in before if order.index(a) > order.index(b)), None)
        if bad is None:
            break
        order.remove(bad[0]); order.insert(order.index(bad[1]), bad[0])
    # softened bonded terms go back into the plain forces (at the end: the reference's indices are not in the document), their new
    # indices to the region's alchemical_bonds / angles / torsions
    from .system import HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce
    softened = {x: dict(bonds=[], angles=[], torsions=[]) for x in order}
    for c in bonded:
        m = re.match(r'lambda_(bonds|angles|torsions)(\w*)\*', compact(c['energy']))
        kind, x = m.group(1), m.group(2)
        if x not in softened:
            raise NotImplementedError('lambda_%s%s of a region without sterics forces (regions: %s)' % (kind, x, sorted(na_s)))
        cls, attr = {'bonds': (HarmonicBondForce, 'bonds'), 'angles': (HarmonicAngleForce, 'angles'), 'torsions': (PeriodicTorsionForce, 'torsions')}[kind]
        plain = [f for f in system.getForces() if isinstance(f, cls)]
        if not plain:
            plain = [cls()]
            system.addForce(plain[0])
        items = [((i, j), p) for (i, j, p) in c['bonds']] if kind == 'bonds' else c['terms']
        for ats, p in items:
            if kind == 'torsions':
                p = (int(p[0]), p[1], p[2])
            getattr(plain[-1], attr).append(tuple(ats) + tuple(p))
            softened[x][kind].append(len(getattr(plain[-1], attr)) - 1)
    regions = []
    ewald = nb.getNonbondedMethod() in (NonbondedForce.Ewald, NonbondedForce.PME)
    exact = ewald and not electro
    for x in order:
        g = na_s[x]['globals']
        fixed_s = compact(aa_s[x]['energy']).endswith('lambda_sterics%s=1.0;' % x)
        fixed_e = x in aa_e and compact(aa_e[x]['energy']).endswith('lambda_electrostatics%s=1.0;' % x)
        regions.append(AlchemicalRegion(alchemical_atoms=atoms[x], alchemical_bonds=softened[x]['bonds'] or None, alchemical_angles=softened[x]['angles'] or None,
                                        alchemical_torsions=softened[x]['torsions'] or None, annihilate_electrostatics=not fixed_e, annihilate_sterics=not fixed_s,
                                        softcore_alpha=g['softcore_alpha'], softcore_a=g['softcore_a'], softcore_b=g['softcore_b'], softcore_c=g['softcore_c'],
                                        softcore_beta=g.get('softcore_beta', 0.0), softcore_d=g.get('softcore_d', 1.0), softcore_e=g.get('softcore_e', 1.0),
                                        softcore_f=g.get('softcore_f', 2.0), name=x[1:] if x else None))
    interactions = set()
    for c in sterics:
        if is_nb(c) and len(c['sfx']) == 2:
            interactions.add(tuple(sorted((order.index(c['sfx'][0]), order.index(c['sfx'][1])))))
    # factory options from an electrostatics pair force
    opts = dict(alchemical_pme_treatment='exact', alchemical_rf_treatment='switched', switch_width=0.1, consistent_exceptions=False)
    if na_e:
        c = next(iter(na_e.values()))
        e = compact(c['energy'])
        if 'erfc(alpha_ewald*' in e:
            opts['alchemical_pme_treatment'] = 'direct-space'
        elif 'k_rf*reff_electrostatics^2' in e:
            opts['alchemical_rf_treatment'] = 'switched' if float(re.search(r'c_rf=([^;]+);', e).group(1)) == 0.0 else 'shifted'
        else:
            opts['alchemical_pme_treatment'] = 'coulomb'
        opts['switch_width'] = round(nb.getCutoffDistance() - float(c['attrs'].get('switchingDistance', nb.getCutoffDistance() - 0.1)), 12)
        eb = next((compact(b['energy']) for b in electro if not is_nb(b)), '')
        opts['consistent_exceptions'] = 'erfc(alpha_ewald*' in eb or 'k_rf*reff_electrostatics^2' in eb
    if rf:
        opts['alchemical_rf_treatment'] = 'switched'
        opts['switch_width'] = round(nb.getCutoffDistance() - float(rf[0]['attrs'].get('switchingDistance', nb.getCutoffDistance() - 0.1)), 12)
    elif nb.getNonbondedMethod() == NonbondedForce.CutoffPeriodic and not na_e:
        opts['alchemical_rf_treatment'] = 'shifted'              # no replaced reaction field in the document
    # ---- the reference NonbondedForce again ---------------------------------------------------------------------------------
    charge = {p[1]: p[2] for p in particle_offsets}
    for x in order:
        for k in atoms[x]:
            sig, eps = na_s[x]['particles'][k]                  # a region's own forces were filled before its atoms were zeroed
            q = na_e[x]['particles'][k][0] if x in na_e else charge.get(k, 0.0)
            nb.setParticleParameters(k, q, sig, eps)
    if rf:                                                       # replace_reaction_field moved ALL charges there
        for k, p in enumerate(rf[0]['particles']):
            if k not in owner:
                q, s_, e_ = nb.getParticleParameters(k)
                nb.setParticleParameters(k, p[0], s_, e_)
    qq_of = {p[1]: p[2] for p in exception_offsets}
    lj_of = {frozenset((i, j)): (p[0], p[1]) for bl in bonds_s.values() for (i, j, p) in bl}
    qq_bond = {frozenset((i, j)): p[0] for bl in bonds_e.values() for (i, j, p) in bl}
    for n, (i, j, qq, s_, e_) in enumerate(list(nb.exceptions)):
        if i in owner or j in owner:
            key = frozenset((i, j))
            sig, eps = lj_of.get(key, (s_, 0.0))
            nb.exceptions[n] = (i, j, qq_of.get(n, qq_bond.get(key, 0.0)), sig, eps)
    if exact and len(order) > 1:
        # the exclusions the factory put between regions that do not interact (:1663-1672) are its own: this package's factory (and the
        # engine) exclude those pairs without listing them
        slot = {x: k for k, x in enumerate(order)}
        nb.exceptions[:] = [e for e in nb.exceptions
                            if not (e[0] in owner and e[1] in owner and owner[e[0]] != owner[e[1]] and (e[2], e[3], e[4]) == (0.0, 1.0, 0.0)
                                    and tuple(sorted((slot[owner[e[0]]], slot[owner[e[1]]]))) not in interactions)]
    first = na_s[order[0]]
    use_lrc = first['attrs'].get('useLongRangeCorrection', '0') not in ('0', 'false')
    if particle_offsets or global_parameters:                    # the factory put it into a lambda_electrostatics group
        nb.setForceGroup(0)
    factory = AbsoluteAlchemicalFactory(disable_alchemical_dispersion_correction=not (use_lrc or not nb.getUseDispersionCorrection()), **opts)
    return factory.create_alchemical_system(system, regions, alchemical_regions_interactions=frozenset(interactions))
