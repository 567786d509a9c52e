"""Langevin splitting integrators as parameter carriers for the device engine.

Mirrors openmmtools/integrators.py: LangevinIntegrator (:1015-1557), VVVRIntegrator (:2125),
BAOABIntegrator (:2152), GeodesicBAOABIntegrator (:2194).  The reference builds an OpenMM
CustomIntegrator step program; here the splitting string is parsed with the same rules
(:1474-1537, sanity check :1337-1402) and handed to libremd_hip.so, whose integrate_chain
kernel executes the V/R/O substeps (csrc/integrate.hip).
"""
import math
from . import unit, constants


class LangevinIntegrator:
    def __init__(self, temperature=298.0 * unit.kelvin, collision_rate=1.0 / unit.picoseconds,
                 timestep=1.0 * unit.femtoseconds, splitting="V R O R V", constraint_tolerance=1e-8,
                 measure_shadow_work=False, measure_heat=False):
        self._temperature = float(temperature)
        self._gamma = float(collision_rate)
        self._timestep = float(timestep)
        self._constraint_tolerance = float(constraint_tolerance)
        self._splitting = splitting
        self._ORV_counts, self._mts, self._force_group_nV = self._parse_splitting_string(splitting)
        self._metropolized_integrator = '{' in splitting                     # integrators.py:1114-1119
        self._measure_heat = bool(measure_heat)
        self._measure_shadow_work = bool(measure_shadow_work) or self._metropolized_integrator

    # ---- reference API ---------------------------------------------------------------
    def getStepSize(self):
        return self._timestep

    def getTemperature(self):
        return self._temperature

    def setTemperature(self, temperature):
        self._temperature = float(temperature)

    def getConstraintTolerance(self):
        return self._constraint_tolerance

    @property
    def kT(self):
        return constants.kB * self._temperature

    @property
    def is_metropolized(self):
        """integrators.py:1304-1307."""
        return self._metropolized_integrator

    @property
    def measure_heat(self):
        return self._measure_heat

    @property
    def measure_shadow_work(self):
        """True for every Metropolized splitting, whatever the constructor flag said (integrators.py:1117-1119)."""
        return self._measure_shadow_work

    @property
    def splitting(self):
        return self._splitting

    @property
    def collision_rate(self):
        return self._gamma

    @property
    def a(self):
        """integrators.py:1142-1143."""
        h = self._timestep / max(1, self._ORV_counts['O'])
        return math.exp(-self._gamma * h)

    @property
    def b(self):
        """integrators.py:1146."""
        h = self._timestep / max(1, self._ORV_counts['O'])
        return math.sqrt(1.0 - math.exp(-2.0 * self._gamma * h))

    # ---- parsing (integrators.py:1337-1402, 1474-1537) --------------------------------------
    @staticmethod
    def _sanity_check(splitting, require_O=True):
        """integrators.py:1319-1402.  Same verdicts and exception types as the reference on every string it treats sensibly
        (tests/golden/splittings_reference.json holds what its own parser says); stricter where the reference lets nonsense
        through by accident: step names must be single letters (it accepts 'OR' and '12' by a substring test), braces must be
        balanced and not nested, no O inside them and no R / V outside them (its two regular expressions only look at the start
        of the string)."""
        tokens = splitting.split(' ')
        depth = 0
        has_braces = '{' in tokens
        for t in tokens:
            if t == '':
                raise ValueError('Invalid step name: splitting has repeated or trailing spaces')
            if t in '{}':
                if t == '{' and '}' not in tokens:
                    raise ValueError('Use of { must be followed by }')                       # :1356-1357
                depth += 1 if t == '{' else -1
                if depth > 1:
                    raise ValueError('There can only be one Metropolized region.')          # :1371-1374
                if depth < 0:      # a '}' in front of its '{': the reference's verdict on 'O } V R V { O' (:1358-1359 through :1376-1402)
                    raise ValueError('Shadow work generating steps found outside the Metropolization block')
                continue
            allowed_characters = "0123456789" + "O" + "R" + "{" + "}" + "V"              # :1336-1340: digits + the dispatch table's keys in its order
            if t[0] not in 'ORV' or (t[0] != 'V' and len(t) > 1):
                raise ValueError("Invalid step name '{}' used; valid step names are {}".format(t, allowed_characters))   # :1358
            if t[0] == 'V' and len(t) > 1 and not (t[1:].isdigit() and int(t[1:]) <= 31):
                # :1343-1350: the reference's "OpenMM only allows up to 32 force groups" is raised inside the try whose except
                # turns every ValueError into this sentence, so this is what a user sees for V32 as well
                raise ValueError("You must use an integer force group")
            if t[0] == 'O' and depth > 0:
                raise ValueError('O steps cannot be inside the Metropolization block')
            if t[0] in 'RV' and has_braces and depth == 0:
                raise ValueError('Shadow work generating steps found outside the Metropolization block')   # :1358-1359
        if depth != 0:
            raise ValueError('Use of { must be followed by }')
        # :1365-1368: the reference asserts that all three kinds of step occur
        assert any(t == 'R' for t in tokens)
        assert any(t[0] == 'V' for t in tokens)
        assert any(t == 'O' for t in tokens) or not require_O

    def _parse_splitting_string(self, splitting_string):
        splitting_string = splitting_string.upper()
        self._sanity_check(splitting_string, require_O=getattr(self, '_REQUIRE_O', True))
        steps = splitting_string.split(' ')
        counts = {s: sum(1 for t in steps if t[0] == s) for s in 'ORV{}'}
        groups = set(t[1:] for t in steps if t[0] == 'V' and len(t) > 1)
        mts = len(groups) > 1
        if mts:                                             # integrators.py:1524-1533
            for t in steps:
                # integrators.py:1527-1529: every V of a multiple-time-step splitting names its force group
                assert not (t[0] == 'V' and len(t) == 1), 'a multiple-time-step splitting must name the force group of every V step'
            return counts, mts, {g: sum(1 for t in steps if t[0] == 'V' and t[1:] == g) for g in groups}
        return counts, mts, {'0': counts['V']}


class VVVRIntegrator(LangevinIntegrator):
    def __init__(self, *args, **kwargs):
        kwargs['splitting'] = "O V R V O"       # integrators.py:2149
        super().__init__(*args, **kwargs)


class BAOABIntegrator(LangevinIntegrator):
    def __init__(self, *args, **kwargs):
        kwargs['splitting'] = "V R O R V"       # integrators.py:2190
        super().__init__(*args, **kwargs)


class GeodesicBAOABIntegrator(LangevinIntegrator):
    def __init__(self, *args, K_r=2, **kwargs):
        kwargs['splitting'] = " ".join(["V"] + ["R"] * K_r + ["O"] + ["R"] * K_r + ["V"])   # :2237-2238
        super().__init__(*args, **kwargs)


class GHMCIntegrator(LangevinIntegrator):
    """integrators.py:2242-2289: generalized hybrid Monte Carlo = the Metropolized splitting "O { V R V } O"."""

    SPLITTING = "O { V R V } O"                 # integrators.py:2286

    def __init__(self, *args, **kwargs):
        kwargs['splitting'] = self.SPLITTING
        super().__init__(*args, **kwargs)


class VelocityVerletIntegrator(LangevinIntegrator):
    """integrators.py:456-498: velocity Verlet with constraints -- the deterministic splitting "V R V" (no thermostat step: the
    collision rate is zero and the temperature unused)."""

    _REQUIRE_O = False

    def __init__(self, timestep=1.0 * unit.femtoseconds, **kwargs):
        kwargs.pop('splitting', None)
        super().__init__(temperature=kwargs.pop('temperature', 298.0 * unit.kelvin), collision_rate=0.0, timestep=timestep,
                         splitting='V R V', **kwargs)


class HMCIntegrator(LangevinIntegrator):
    """integrators.py:885-1010: hybrid Monte Carlo -- velocities redrawn from the Maxwell-Boltzmann distribution, ``nsteps``
    velocity Verlet steps, one Metropolis test: one pass of "O { (V R V)^nsteps }" at nsteps * timestep with the collision
    rate at which the O step forgets the old velocities completely (mcmc.HMCMove runs the same program)."""

    def __init__(self, temperature=298.0 * unit.kelvin, nsteps=10, timestep=1.0 * unit.femtoseconds, **kwargs):
        n = int(nsteps)
        if n < 1:
            raise ValueError('HMCIntegrator needs at least one step per trajectory')
        kwargs.pop('splitting', None)
        kwargs.pop('collision_rate', None)
        self.nsteps = n
        self.hmc_timestep = float(unit.to_md(timestep)) if hasattr(unit, 'to_md') else float(timestep)
        super().__init__(temperature=temperature, collision_rate=1.0e12, timestep=self.hmc_timestep * n,
                         splitting='O {' + ' V R V' * n + ' }', **kwargs)

    @property
    def acceptance_rate(self):
        """:1003-1006 reads the integrator's counters; here they live on the engine (remd_get_work)."""
        raise AttributeError('acceptance statistics are per replica on the engine: engine.get_work()')
