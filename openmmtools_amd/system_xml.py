"""``openmm.System`` XML in and out of the engine's ``System`` shim, without OpenMM.

SURVEY 8(f) rank 4 asks for an adapter from real ``openmm.System`` XML: the reference stores every thermodynamic
state's System as ``XmlSerializer.serialize(system)`` (states.py:2757-2790 ``_standardize_and_hash`` /
multistatereporter.py:1377-1470), so a user's existing system.xml is the natural way in.  The element and attribute names
below restate OpenMM's serialization proxies (SystemProxy, HarmonicBondForceProxy, HarmonicAngleForceProxy,
PeriodicTorsionForceProxy, NonbondedForceProxy, CustomExternalForceProxy, CMMotionRemoverProxy,
MonteCarloBarostatProxy; OpenMM 8 writes ``version`` 1-4 of them).  **External knowledge — OpenMM is not in
/root/reference and not installed here**: the reader is written to be tolerant (unknown attributes are ignored,
optional blocks may be missing).  Pinned by round trips through the writer in this module, a hand-written document
(tests/test_system_xml_cpu.py) and — round 3 — **a document written by OpenMM 7.7 itself**: the System of
AlanineDipeptideExplicit stored in the reference's data/reporter-examples/alanine_dipeptide_legacy.nc
(tests/golden/openmm_alanine_fixture.npz, tests/test_openmm_fixture.py) parses to exactly the description this
package builds from the Amber files.

Round 4: a System marked by ``alchemy.AbsoluteAlchemicalFactory`` is written as the force set the reference's factory builds
(NonbondedForce offsets, soft-core CustomNonbondedForces / CustomBondForces: ``_alchemical_xml.py``) and such a document is
read back into a marked System.  Other forces outside the hot path's scope (GBSA, Custom*Force with any other expression,
other NonbondedForce parameter offsets) raise ``NotImplementedError`` naming the force, like ``system_to_desc`` does.
"""
import xml.etree.ElementTree as ET

from .system import (System, HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce, NonbondedForce, GBSAOBCForce,
                     CustomExternalForce, CMMotionRemover)


def _f(x):
    return repr(float(x))


# ---- writer ------------------------------------------------------------------------------------------------------
def _emit_force(forces, f, force_group=None, particles=None, exceptions=None, global_parameters=(), particle_offsets=(),
                exception_offsets=()):
    """One force of the System shim as a <Force> element.  The keyword arguments are the NonbondedForce of an alchemically
    modified System (_alchemical_xml.py): replaced particle / exception tables, the global parameter and its offsets."""
    name = type(f).__name__
    common = dict(forceGroup=str(f.getForceGroup() if force_group is None else force_group), name=name, type=name)
    if isinstance(f, HarmonicBondForce):
        e = ET.SubElement(forces, 'Force', dict(common, usesPeriodic='0', version='2'))
        b = ET.SubElement(e, 'Bonds')
        for (i, j, d, k) in f.bonds:
            ET.SubElement(b, 'Bond', dict(d=_f(d), k=_f(k), p1=str(i), p2=str(j)))
    elif isinstance(f, HarmonicAngleForce):
        e = ET.SubElement(forces, 'Force', dict(common, usesPeriodic='0', version='2'))
        b = ET.SubElement(e, 'Angles')
        for (i, j, k, a, kf) in f.angles:
            ET.SubElement(b, 'Angle', dict(a=_f(a), k=_f(kf), p1=str(i), p2=str(j), p3=str(k)))
    elif isinstance(f, PeriodicTorsionForce):
        e = ET.SubElement(forces, 'Force', dict(common, usesPeriodic='0', version='2'))
        b = ET.SubElement(e, 'Torsions')
        for (i, j, k, l, per, ph, kf) in f.torsions:
            ET.SubElement(b, 'Torsion', dict(k=_f(kf), p1=str(i), p2=str(j), p3=str(k), p4=str(l),
                                             periodicity=str(int(per)), phase=_f(ph)))
    elif isinstance(f, NonbondedForce):
        alpha, nx, ny, nz = f._pme_params if f._pme_params else (0.0, 0, 0, 0)
        e = ET.SubElement(forces, 'Force', dict(
            common, alpha=_f(alpha), cutoff=_f(f.getCutoffDistance()),
            dispersionCorrection=str(int(f.getUseDispersionCorrection())), ewaldTolerance=_f(f.getEwaldErrorTolerance()),
            exceptionsUsePeriodic='0', includeDirectSpace='1', ljAlpha='0', ljnx='0', ljny='0', ljnz='0',
            method=str(f.getNonbondedMethod()), nx=str(nx), ny=str(ny), nz=str(nz), recipForceGroup='-1',
            rfDielectric=_f(f.getReactionFieldDielectric()), switchingDistance=_f(f.getSwitchingDistance()),
            useSwitchingFunction=str(int(f.getUseSwitchingFunction())), version='4'))
        g = ET.SubElement(e, 'GlobalParameters')
        for (pname, default) in global_parameters:
            ET.SubElement(g, 'Parameter', dict(default=_f(default), name=pname))
        g = ET.SubElement(e, 'ParticleOffsets')
        for (pname, idx, q, sig, eps) in particle_offsets:
            ET.SubElement(g, 'Offset', dict(eps=_f(eps), parameter=pname, particle=str(idx), q=_f(q), sig=_f(sig)))
        g = ET.SubElement(e, 'ExceptionOffsets')
        for (pname, idx, q, sig, eps) in exception_offsets:
            ET.SubElement(g, 'Offset', dict(eps=_f(eps), exception=str(idx), parameter=pname, q=_f(q), sig=_f(sig)))
        b = ET.SubElement(e, 'Particles')
        for (q, sig, eps) in (f.particles if particles is None else particles):
            ET.SubElement(b, 'Particle', dict(eps=_f(eps), q=_f(q), sig=_f(sig)))
        b = ET.SubElement(e, 'Exceptions')
        for (i, j, qq, sig, eps) in (f.exceptions if exceptions is None else exceptions):
            ET.SubElement(b, 'Exception', dict(eps=_f(eps), p1=str(i), p2=str(j), q=_f(qq), sig=_f(sig)))
    elif isinstance(f, CustomExternalForce):
        e = ET.SubElement(forces, 'Force', dict(common, energy=f.energy_expression, version='1'))
        ET.SubElement(e, 'PerParticleParameters')
        g = ET.SubElement(e, 'GlobalParameters')
        for k, v in f.globals.items():
            ET.SubElement(g, 'Parameter', dict(default=_f(v), name=k))
        b = ET.SubElement(e, 'Particles')
        for idx in f.particles:
            ET.SubElement(b, 'Particle', dict(index=str(idx)))
    elif isinstance(f, CMMotionRemover):
        ET.SubElement(forces, 'Force', dict(common, frequency=str(f.getFrequency()), version='1'))
    elif isinstance(f, GBSAOBCForce):
        # OpenMM's GBSAOBCForceProxy (external knowledge, unpinned like the custom forces of _alchemical_xml.py)
        e = ET.SubElement(forces, 'Force', dict(common, method=str(f.getNonbondedMethod()), cutoff=_f(1.0), soluteDielectric=_f(f.getSoluteDielectric()),
                                                solventDielectric=_f(f.getSolventDielectric()), surfaceAreaEnergy=_f(f.getSurfaceAreaEnergy()), version='2'))
        b = ET.SubElement(e, 'Particles')
        for (q, r, sc) in f.particles:
            ET.SubElement(b, 'Particle', dict(q=_f(q), r=_f(r), scale=_f(sc)))
    else:
        raise NotImplementedError('unsupported force %r' % name)


def to_xml(system, pressure=None, temperature=None, barostat_frequency=25):
    """Serialise ``system`` (and, when ``pressure`` is given, a MonteCarloBarostat force in bar / kelvin, which is how
    the reference carries the pressure of an NPT ThermodynamicState, states.py:1020-1068)."""
    root = ET.Element('System', dict(openmmVersion='8.0', type='System', version='1'))
    box = ET.SubElement(root, 'PeriodicBoxVectors')
    for tag, v in zip('ABC', system.getDefaultPeriodicBoxVectors()):
        ET.SubElement(box, tag, dict(x=_f(v[0]), y=_f(v[1]), z=_f(v[2])))
    parts = ET.SubElement(root, 'Particles')
    for i in range(system.getNumParticles()):
        ET.SubElement(parts, 'Particle', dict(mass=_f(system.getParticleMass(i))))
    cons = ET.SubElement(root, 'Constraints')
    for i in range(system.getNumConstraints()):
        p, q, d = system.getConstraintParameters(i)
        ET.SubElement(cons, 'Constraint', dict(d=_f(d), p1=str(p), p2=str(q)))
    forces = ET.SubElement(root, 'Forces')
    if getattr(system, 'alchemical_regions', None) is not None:
        # ... in the general-regions mode (several / named regions, soft-core electrostatics, the non-exact PME treatments)
        from . import _alchemical_xml
        _alchemical_xml.emit_region_forces(forces, system, lambda f, **kw: _emit_force(forces, f, **kw))
    elif getattr(system, 'alchemical_region', None) is not None:
        # a System marked by alchemy.AbsoluteAlchemicalFactory is written as the force set the reference's factory builds
        from . import _alchemical_xml
        _alchemical_xml.emit_alchemical_forces(forces, system, lambda f, **kw: _emit_force(forces, f, **kw))
    else:
        for f in system.getForces():
            _emit_force(forces, f)
    if pressure is not None:
        ET.SubElement(forces, 'Force', dict(forceGroup='0', name='MonteCarloBarostat', type='MonteCarloBarostat',
                                            pressure=_f(pressure), temperature=_f(temperature if temperature is not None else 300.0),
                                            frequency=str(int(barostat_frequency)), randomSeed='0', version='1'))
    ET.indent(root, space='\t')
    return '<?xml version="1.0" ?>\n' + ET.tostring(root, encoding='unicode') + '\n'


# ---- reader ------------------------------------------------------------------------------------------------------
def _children(elem, block, item):
    b = elem.find(block)
    return [] if b is None else b.findall(item)


def _nonempty(elem, block):
    b = elem.find(block)
    return b is not None and len(list(b)) > 0


def from_xml(text_or_path):
    """Parse an OpenMM System XML document (a string starting with '<' or a file path).

    Returns ``(system, barostat)`` with ``barostat`` = None or a dict(pressure [bar], temperature [K], frequency) taken
    from a MonteCarloBarostat force, which the engine carries on the ThermodynamicState instead of in the System."""
    text = text_or_path
    if not str(text_or_path).lstrip().startswith('<'):
        with open(text_or_path) as fh:
            text = fh.read()
    root = ET.fromstring(text)
    if root.tag != 'System':
        raise ValueError('not an OpenMM System document (root element %r)' % root.tag)
    s = System()
    box = root.find('PeriodicBoxVectors')
    if box is not None:
        vec = [tuple(float(box.find(t).get(c)) for c in 'xyz') for t in 'ABC']
        s.setDefaultPeriodicBoxVectors(*vec)
    for p in _children(root, 'Particles', 'Particle'):
        s.addParticle(float(p.get('mass')))
    for c in _children(root, 'Constraints', 'Constraint'):
        s.addConstraint(int(c.get('p1')), int(c.get('p2')), float(c.get('d')))
    barostat = None
    customs, alch_nb, nb_force = [], None, None
    for e in _children(root, 'Forces', 'Force'):
        kind = e.get('type')
        if kind == 'HarmonicBondForce':
            f = HarmonicBondForce()
            for b in _children(e, 'Bonds', 'Bond'):
                f.addBond(int(b.get('p1')), int(b.get('p2')), float(b.get('d')), float(b.get('k')))
        elif kind == 'HarmonicAngleForce':
            f = HarmonicAngleForce()
            for b in _children(e, 'Angles', 'Angle'):
                f.addAngle(int(b.get('p1')), int(b.get('p2')), int(b.get('p3')), float(b.get('a')), float(b.get('k')))
        elif kind == 'PeriodicTorsionForce':
            f = PeriodicTorsionForce()
            for b in _children(e, 'Torsions', 'Torsion'):
                f.addTorsion(int(b.get('p1')), int(b.get('p2')), int(b.get('p3')), int(b.get('p4')),
                             int(b.get('periodicity')), float(b.get('phase')), float(b.get('k')))
        elif kind == 'NonbondedForce':
            # parameter offsets: understood as the alchemical factory's lambda_electrostatics (checked after the loop)
            nb_globals = {g.get('name'): float(g.get('default')) for g in _children(e, 'GlobalParameters', 'Parameter')}
            offsets = [[(o.get('parameter'), int(o.get(key)), float(o.get('q')), float(o.get('sig')), float(o.get('eps')))
                        for o in _children(e, block, 'Offset')] for block, key in (('ParticleOffsets', 'particle'), ('ExceptionOffsets', 'exception'))]
            if nb_globals or offsets[0] or offsets[1]:
                if alch_nb is not None:
                    raise NotImplementedError('parameter offsets on more than one NonbondedForce')
                alch_nb = (nb_globals, offsets[0], offsets[1])
            f = NonbondedForce()
            nb_force = f
            f.setNonbondedMethod(int(e.get('method')))
            f.setCutoffDistance(float(e.get('cutoff')))
            f.setUseSwitchingFunction(bool(int(e.get('useSwitchingFunction', '0'))))
            f.setSwitchingDistance(float(e.get('switchingDistance', '-1')))
            f.setUseDispersionCorrection(bool(int(e.get('dispersionCorrection', '1'))))
            f.setReactionFieldDielectric(float(e.get('rfDielectric', '78.3')))
            f.setEwaldErrorTolerance(float(e.get('ewaldTolerance', '0.0005')))
            alpha, nx = float(e.get('alpha', '0')), int(e.get('nx', '0'))
            if alpha != 0.0 or nx != 0:
                f.setPMEParameters(alpha, nx, int(e.get('ny', '0')), int(e.get('nz', '0')))
            for b in _children(e, 'Particles', 'Particle'):
                f.addParticle(float(b.get('q')), float(b.get('sig')), float(b.get('eps')))
            for b in _children(e, 'Exceptions', 'Exception'):
                f.addException(int(b.get('p1')), int(b.get('p2')), float(b.get('q')), float(b.get('sig')), float(b.get('eps')))
        elif kind in ('CustomNonbondedForce', 'CustomBondForce', 'CustomAngleForce', 'CustomTorsionForce'):
            from . import _alchemical_xml
            customs.append(_alchemical_xml.parse_custom(e))        # only as the pieces of an alchemically modified System
            continue
        elif kind == 'CustomGBForce' and e.find('ComputedValues') is not None:
            from . import _alchemical_xml
            customs.append(_alchemical_xml.parse_custom_gb(e))     # the alchemical factory's GBSA (alchemy.py:2172-2225)
            continue
        elif kind == 'CustomExternalForce':
            if _nonempty(e, 'PerParticleParameters'):
                raise NotImplementedError('CustomExternalForce with per-particle parameters')
            f = CustomExternalForce(e.get('energy'))          # raises for anything but the harmonic-oscillator expression
            for g in _children(e, 'GlobalParameters', 'Parameter'):
                f.addGlobalParameter(g.get('name'), float(g.get('default')))
            for b in _children(e, 'Particles', 'Particle'):
                f.addParticle(int(b.get('index')), [])
        elif kind == 'CMMotionRemover':
            f = CMMotionRemover(int(e.get('frequency', '1')))
        elif kind == 'GBSAOBCForce':
            f = GBSAOBCForce()
            f.setNonbondedMethod(int(e.get('method', '0')))
            f.setSoluteDielectric(float(e.get('soluteDielectric', '1'))); f.setSolventDielectric(float(e.get('solventDielectric', '78.5')))
            f.setSurfaceAreaEnergy(float(e.get('surfaceAreaEnergy', '2.25936')))
            for b in _children(e, 'Particles', 'Particle'):
                f.addParticle(float(b.get('q')), float(b.get('r')), float(b.get('scale')))
        elif kind == 'MonteCarloBarostat':
            barostat = dict(pressure=float(e.get('pressure')), temperature=float(e.get('temperature', '300')),
                            frequency=int(e.get('frequency', '25')))
            continue
        elif kind == 'AndersenThermostat':
            # the reference's *standard system* keeps an AndersenThermostat as the marker of "this state has a
            # thermostat" (states.py:1447-1490, 1100-1180); its temperature lives on the ThermodynamicState and the
            # Langevin integrator does the thermostatting, so the force itself carries nothing for the engine
            continue
        else:
            raise NotImplementedError('unsupported OpenMM force %s' % kind)
        f.setForceGroup(int(e.get('forceGroup', '0')))
        s.addForce(f)
    if customs or alch_nb is not None:
        from . import _alchemical_xml
        if nb_force is None:
            raise NotImplementedError('custom forces without a NonbondedForce (%s)' % customs[0]['type'])
        g, po, eo = alch_nb if alch_nb is not None else ({}, [], [])
        s = _alchemical_xml.rebuild_marked_system(s, nb_force, g, po, eo, customs)
    return s, barostat
