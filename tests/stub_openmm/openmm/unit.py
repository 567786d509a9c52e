"""Minimal ``openmm.unit`` look-alike (TEST INFRASTRUCTURE ONLY): Unit = factor into the md unit system plus a dimension
signature; Quantity = value x Unit with value_in_unit / value_in_unit_system; ``md_unit_system`` sentinel."""
import numpy as np


class Unit:
    def __init__(self, factor, dims):
        self.factor, self.dims = float(factor), dict(dims)

    def _combine(self, other, sign):
        d = dict(self.dims)
        for k, v in other.dims.items():
            d[k] = d.get(k, 0) + sign * v
        return {k: v for k, v in d.items() if v != 0}

    def __mul__(self, other):
        if isinstance(other, Unit):
            return Unit(self.factor * other.factor, self._combine(other, +1))
        return Quantity(other, self)
    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return Unit(self.factor / other.factor, self._combine(other, -1))
        return Quantity(1.0 / other, self)

    def __rtruediv__(self, other):
        return Quantity(other, Unit(1.0 / self.factor, {k: -v for k, v in self.dims.items()}))

    def __pow__(self, n):
        return Unit(self.factor ** n, {k: v * n for k, v in self.dims.items()})


class UnitSystem:
    pass


md_unit_system = UnitSystem()


class Quantity:
    def __init__(self, value, unit):
        self._value, self.unit = value, unit

    def value_in_unit(self, unit):
        if unit.dims != self.unit.dims:
            raise TypeError('incompatible units')
        return self._scaled(self.unit.factor / unit.factor)

    def value_in_unit_system(self, system):
        assert system is md_unit_system
        return self._scaled(self.unit.factor)

    def _scaled(self, f):
        v = self._value
        if isinstance(v, (list, tuple)):
            return type(v)(*[x * f for x in v]) if type(v).__name__ == 'Vec3' else [Quantity(x, Unit(1, {}))._scaled(f) if isinstance(x, (list, tuple)) else x * f for x in v]
        return np.asarray(v) * f if isinstance(v, np.ndarray) else v * f

    def __mul__(self, other):
        if isinstance(other, Unit):
            return Quantity(self._value, self.unit * other)
        return Quantity(self._value * other, self.unit)
    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return Quantity(self._value, self.unit / other)
        return Quantity(self._value / other, self.unit)


def _q(value, unit):
    """value already a Quantity -> as is; plain number -> Quantity in `unit` (what OpenMM's setters accept)."""
    return value if isinstance(value, Quantity) else Quantity(value, unit)


def is_quantity(x):
    return isinstance(x, Quantity)


nanometer = nanometers = Unit(1.0, {'L': 1})
angstrom = angstroms = Unit(0.1, {'L': 1})
picosecond = picoseconds = Unit(1.0, {'T': 1})
femtosecond = femtoseconds = Unit(1e-3, {'T': 1})
amu = dalton = daltons = Unit(1.0, {'M': 1})
kelvin = kelvins = Unit(1.0, {'K': 1})
elementary_charge = elementary_charges = Unit(1.0, {'Q': 1})
radian = radians = Unit(1.0, {})
degree = degrees = Unit(0.017453292519943295, {})
kilojoule_per_mole = kilojoules_per_mole = Unit(1.0, {'M': 1, 'L': 2, 'T': -2})
kilocalorie_per_mole = kilocalories_per_mole = Unit(4.184, {'M': 1, 'L': 2, 'T': -2})
bar = bars = Unit(1.0e5 * 1.0e-27 * 6.02214076e23 * 1.0e-3, {'M': 1, 'L': -1, 'T': -2})
atmosphere = atmospheres = Unit(1.01325 * bar.factor, {'M': 1, 'L': -1, 'T': -2})
