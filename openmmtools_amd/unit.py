"""Plain-float stand-in for ``openmm.unit`` in OpenMM's md_unit_system.

Every unit is a float multiplier into (nm, ps, amu, kJ/mol, K, e), so ``300*unit.kelvin`` or
``2.0*unit.femtoseconds`` produce the numbers the engine consumes.  The reference wraps
everything in ``openmm.unit.Quantity`` (e.g. openmmtools/states.py:1908-1917); OpenMM is not
importable where this package is built, and the hot path only ever needs md-unit floats.
"""
# length
nanometer = nanometers = 1.0
angstrom = angstroms = 0.1
# time
picosecond = picoseconds = 1.0
femtosecond = femtoseconds = 1.0e-3
nanosecond = nanoseconds = 1.0e3
# mass / charge / temperature
amu = dalton = daltons = 1.0
elementary_charge = elementary_charges = 1.0
kelvin = kelvins = 1.0
# energy
kilojoule_per_mole = kilojoules_per_mole = 1.0
kilocalorie_per_mole = kilocalories_per_mole = 4.184
# pressure (times nm^3 gives kJ/mol):  1 bar nm^3 = 1e5 Pa * 1e-27 m^3 * N_A / 1000
AVOGADRO_CONSTANT_NA = 6.02214076e23
BOLTZMANN_CONSTANT_kB = 1.380649e-23 * 1.0e-3      # kJ/K per particle
bar = bars = 1.0e5 * 1.0e-27 * AVOGADRO_CONSTANT_NA * 1.0e-3
atmosphere = atmospheres = 1.01325 * bar
# angles
radian = radians = 1.0
degree = degrees = 0.017453292519943295


def to_md(value):
    """A plain number / array in OpenMM's md_unit_system (nm, ps, amu, kJ/mol, K, e; pressure kJ/mol/nm^3) from either
    a plain number (taken as already being in those units: the floats this module's unit constants produce) or anything
    that behaves like ``openmm.unit.Quantity`` -- i.e. offers ``value_in_unit_system`` -- which is what a caller that
    builds its states with the reference's own objects passes (openmmtools/states.py:1908-1917 strips units the same
    way).  The unit system object is looked up next to the Quantity's class, so no import of openmm is needed here."""
    if hasattr(value, 'value_in_unit_system'):
        import importlib
        import sys
        mod = sys.modules.get(type(value).__module__)
        md = None
        for name in (type(value).__module__, type(value).__module__.rsplit('.', 1)[0], 'openmm.unit', 'simtk.unit'):
            try:
                m = sys.modules.get(name) or importlib.import_module(name)
            except ImportError:
                continue
            md = getattr(m, 'md_unit_system', None)
            if md is not None:
                break
        if md is None:
            raise TypeError('cannot find md_unit_system for a %s' % type(value).__name__)
        return value.value_in_unit_system(md)
    return value
