"""A MINIMAL STAND-IN for the ``openmm`` package (TEST INFRASTRUCTURE ONLY; OpenMM is not installable here).

It mimics exactly the part of OpenMM's Python API that openmmtools_amd.system.from_openmm and the boundary code of
openmmtools_amd.states / mcmc touch: ``System`` and the five force classes of the benchmark systems with their
unit-carrying getters, ``Vec3`` and ``openmm.unit`` (Quantity with value_in_unit / value_in_unit_system, the md unit
system, unit arithmetic).  The getters return ``Quantity`` objects in OpenMM's conventions (nm, kJ/mol, radians, e), as
the real ones do.  tests/test_integration_stub.py puts this directory on sys.path.
"""
from . import unit
from .unit import Quantity


class Vec3(tuple):
    def __new__(cls, x, y, z):
        return tuple.__new__(cls, (x, y, z))


class Force:
    pass


class HarmonicBondForce(Force):
    def __init__(self):
        self._b = []

    def addBond(self, p, q, r0, k):
        self._b.append((p, q, unit._q(r0, unit.nanometer), unit._q(k, unit.kilojoule_per_mole / unit.nanometer ** 2)))

    def getNumBonds(self):
        return len(self._b)

    def getBondParameters(self, i):
        return self._b[i]


class HarmonicAngleForce(Force):
    def __init__(self):
        self._a = []

    def addAngle(self, p, q, r, th, k):
        self._a.append((p, q, r, unit._q(th, unit.radian), unit._q(k, unit.kilojoule_per_mole / unit.radian ** 2)))

    def getNumAngles(self):
        return len(self._a)

    def getAngleParameters(self, i):
        return self._a[i]


class PeriodicTorsionForce(Force):
    def __init__(self):
        self._t = []

    def addTorsion(self, p, q, r, s, per, phase, k):
        self._t.append((p, q, r, s, per, unit._q(phase, unit.radian), unit._q(k, unit.kilojoule_per_mole)))

    def getNumTorsions(self):
        return len(self._t)

    def getTorsionParameters(self, i):
        return self._t[i]


class NonbondedForce(Force):
    NoCutoff, CutoffNonPeriodic, CutoffPeriodic, Ewald, PME, LJPME = range(6)

    def __init__(self):
        self._p, self._e = [], []
        self._method, self._cut, self._sw, self._swd, self._disp, self._rf, self._tol = 0, 1.0, False, -1.0, True, 78.3, 5e-4

    def addParticle(self, q, sig, eps):
        self._p.append((unit._q(q, unit.elementary_charge), unit._q(sig, unit.nanometer), unit._q(eps, unit.kilojoule_per_mole)))

    def addException(self, p, q, qq, sig, eps):
        self._e.append((p, q, unit._q(qq, unit.elementary_charge ** 2), unit._q(sig, unit.nanometer), unit._q(eps, unit.kilojoule_per_mole)))

    def getNumParticles(self):
        return len(self._p)

    def getParticleParameters(self, i):
        return self._p[i]

    def getNumExceptions(self):
        return len(self._e)

    def getExceptionParameters(self, i):
        return self._e[i]

    def setNonbondedMethod(self, m): self._method = m
    def getNonbondedMethod(self): return self._method
    def setCutoffDistance(self, d): self._cut = unit._q(d, unit.nanometer)
    def getCutoffDistance(self): return unit._q(self._cut, unit.nanometer)
    def setUseSwitchingFunction(self, b): self._sw = bool(b)
    def getUseSwitchingFunction(self): return self._sw
    def setSwitchingDistance(self, d): self._swd = unit._q(d, unit.nanometer)
    def getSwitchingDistance(self): return unit._q(self._swd, unit.nanometer)
    def setUseDispersionCorrection(self, b): self._disp = bool(b)
    def getUseDispersionCorrection(self): return self._disp
    def setReactionFieldDielectric(self, e): self._rf = float(e)
    def getReactionFieldDielectric(self): return self._rf
    def setEwaldErrorTolerance(self, t): self._tol = float(t)
    def getEwaldErrorTolerance(self): return self._tol


class CMMotionRemover(Force):
    def __init__(self, frequency=1):
        self._f = int(frequency)

    def getFrequency(self):
        return self._f


class System:
    def __init__(self):
        self._m, self._c, self._f = [], [], []
        self._box = [unit._q(Vec3(2, 0, 0), unit.nanometer), unit._q(Vec3(0, 2, 0), unit.nanometer), unit._q(Vec3(0, 0, 2), unit.nanometer)]

    def addParticle(self, m):
        self._m.append(unit._q(m, unit.amu))
        return len(self._m) - 1

    def getNumParticles(self): return len(self._m)
    def getParticleMass(self, i): return self._m[i]
    def addConstraint(self, p, q, d): self._c.append((p, q, unit._q(d, unit.nanometer)))
    def getNumConstraints(self): return len(self._c)
    def getConstraintParameters(self, i): return self._c[i]
    def addForce(self, f): self._f.append(f); return len(self._f) - 1
    def getForces(self): return list(self._f)
    def getNumForces(self): return len(self._f)
    def setDefaultPeriodicBoxVectors(self, a, b, c): self._box = [unit._q(v, unit.nanometer) for v in (a, b, c)]
    def getDefaultPeriodicBoxVectors(self): return list(self._box)
    def usesPeriodicBoundaryConditions(self):
        return any(isinstance(f, NonbondedForce) and f.getNonbondedMethod() >= NonbondedForce.CutoffPeriodic for f in self._f)
