"""Physical constants in md units (mirrors openmmtools/constants.py:1-21)."""
from math import pi
from . import unit

# kB = BOLTZMANN_CONSTANT_kB * AVOGADRO_CONSTANT_NA  (constants.py:7)  [kJ/mol/K]
kB = unit.BOLTZMANN_CONSTANT_kB * unit.AVOGADRO_CONSTANT_NA

# Coulomb constant in kJ/mol nm / e^2 (constants.py:12-14)
E_CHARGE = 1.602176634e-19
EPSILON0 = 1e-6 * 8.8541878128e-12 / (unit.AVOGADRO_CONSTANT_NA * E_CHARGE ** 2)
ONE_4PI_EPS0 = 1.0 / (4.0 * pi * EPSILON0)
