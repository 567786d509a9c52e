"""ThermodynamicState / SamplerState: the input contract of the replica-exchange engine.

Mirrors the parts of openmmtools/states.py the hot path touches:
  ThermodynamicState   states.py:331-1930  (system, temperature, pressure, beta, reduced potential
                       algebra :1908-1917, compatibility :994-1050)
  SamplerState         states.py:1933-2521 (positions, velocities, box_vectors, cached energies)
  CompoundThermodynamicState + GlobalParameterState (lambda_sterics / lambda_electrostatics)
                       states.py:2524-3046, alchemy.AlchemicalState alchemy.py:86-410
Quantities are md-unit floats / numpy arrays (see unit.py).
"""
import copy
import numpy as np
from . import constants
from .unit import to_md


class _CodedError(Exception):
    """The reference's state errors carry a `code` (class constants, numbered in declaration order) and take their message from a table
    indexed by it (states.py:300-337, 366-377): user code tests `err.code == ThermodynamicsError.NO_THERMOSTAT`."""
    _table = ()

    def __init__(self, code, *args):
        super().__init__(self.error_messages[code].format(*args))
        self.code = code

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        cls.error_messages = {}
        for number, (name, message) in enumerate(cls._table):
            setattr(cls, name, number)
            cls.error_messages[number] = message


class ThermodynamicsError(_CodedError):
    """states.py:272-337 (names, numbers and messages are the reference's: tests/test_signatures.py holds them to its source)."""
    _table = (('MULTIPLE_THERMOSTATS', 'System has multiple thermostats.'),
              ('NO_THERMOSTAT', 'System does not have a thermostat specifying the temperature.'),
              ('NONE_TEMPERATURE', 'Cannot set temperature of the thermodynamic state to None.'),
              ('INCONSISTENT_THERMOSTAT', 'System thermostat is inconsistent with thermodynamic state.'),
              ('MULTIPLE_BAROSTATS', 'System has multiple barostats.'),
              ('NO_BAROSTAT', 'System does not have a barostat specifying the pressure.'),
              ('UNSUPPORTED_BAROSTAT', 'Found unsupported barostat {} in system.'),
              ('UNSUPPORTED_ANISOTROPIC_BAROSTAT', 'MonteCarloAnisotropicBarostat is only supported if the pressure along all scaled axes is the same.'),
              ('SURFACE_TENSION_NOT_SUPPORTED', 'Surface tension can only be set for states that have a system with a MonteCarloMembraneBarostat.'),
              ('INCONSISTENT_BAROSTAT', 'System barostat is inconsistent with thermodynamic state.'),
              ('BAROSTATED_NONPERIODIC', 'Non-periodic systems cannot have a barostat.'),
              ('INCONSISTENT_INTEGRATOR', 'Integrator is coupled to a heat bath at a different temperature.'),
              ('INCOMPATIBLE_SAMPLER_STATE', 'The sampler state has a different number of particles.'),
              ('INCOMPATIBLE_ENSEMBLE', 'Cannot apply to a context in a different thermodynamic ensemble.'))


class SamplerStateError(_CodedError):
    """states.py:340-382."""
    _table = (('INCONSISTENT_VELOCITIES', 'Velocities have different length than positions.'),
              ('INCONSISTENT_POSITIONS', 'Specified positions with inconsistent number of particles.'))


def create_thermodynamic_state_protocol(system, protocol, constants=None, composable_states=None):
    """states.py:39-141: one state per entry of the protocol -- ``protocol`` maps attribute names (temperature, pressure,
    lambda_sterics, ...) to equally long value lists, ``constants`` to values shared by every state; ``system`` is a System
    (then a temperature must be among them) or a ThermodynamicState; ``composable_states`` wrap it into a
    CompoundThermodynamicState first."""
    protocol = dict(protocol)
    if len(protocol) == 0:
        raise ValueError('No protocol has been specified.')
    lengths = {len(v) for v in protocol.values()}
    if len(lengths) != 1:
        raise ValueError('The protocol parameter values have different lengths!\n{}'.format(protocol))
    n = lengths.pop()
    constants_ = dict(constants or {})
    if set(constants_) & set(protocol):
        raise ValueError('Some parameters have been specified both in constants and protocol.')
    for key, value in constants_.items():
        protocol[key] = [value] * n
    if isinstance(system, ThermodynamicState):
        thermo_state = system
    else:
        if 'temperature' in constants_:
            temperature = constants_['temperature']
        elif 'temperature' in protocol:
            temperature = protocol['temperature'][0]
        else:
            raise ValueError('If a System is passed the list of constants must specify the temperature.')
        thermo_state = ThermodynamicState(system, temperature=temperature)
    if composable_states is not None:
        if not isinstance(composable_states, (list, tuple)):
            composable_states = [composable_states]
        thermo_state = CompoundThermodynamicState(thermo_state, composable_states)
    out = [copy.deepcopy(thermo_state) for _ in range(n)]
    for k, state in enumerate(out):
        for key, values in protocol.items():
            if not hasattr(state, key):
                raise AttributeError('{} object does not have protocol attribute {}'.format(type(state), key))
            setattr(state, key, values[k])
    return out


class MonteCarloBarostatSettings:
    """The accessors of openmm.MonteCarloBarostat the reference's code and tests read from ``ThermodynamicState.barostat``."""

    def __init__(self, pressure, temperature, frequency):
        self._pressure, self._temperature, self._frequency = float(pressure), float(temperature), int(frequency)

    def getDefaultPressure(self):
        return self._pressure                     # kJ/mol/nm^3 (``pressure / unit.bar`` gives bar)

    def getDefaultTemperature(self):
        return self._temperature

    def getFrequency(self):
        return self._frequency


class ThermodynamicState:
    def __init__(self, system, temperature=None, pressure=None, surface_tension=None):
        """states.py:507-508, 1314-1353.  A System here carries no thermostat force a temperature could be read from and no membrane
        barostat: no temperature is the reference's NO_THERMOSTAT, a surface tension its INCOMPATIBLE_ENSEMBLE."""
        self._system = system
        if surface_tension is not None:
            raise ThermodynamicsError(ThermodynamicsError.INCOMPATIBLE_ENSEMBLE)          # states.py:1330-1331
        if temperature is None:
            raise ThermodynamicsError(ThermodynamicsError.NO_THERMOSTAT)                  # states.py:1336-1339
        self.temperature = temperature
        # NPT: the reference adds an openmm.MonteCarloBarostat (frequency 25) to the System (states.py:1177-1181); here the
        # pressure (kJ/mol/nm^3, i.e. `p * unit.bar`) and the frequency are handed to the engine's barostat
        self.pressure = pressure                                                 # Quantity -> kJ/mol/nm^3 (md units)
        self.barostat_frequency = 25

    @property
    def pressure(self):
        return self._pressure

    @pressure.setter
    def pressure(self, value):
        """states.py:680-703."""
        if value is not None and not self._system.usesPeriodicBoundaryConditions():
            raise ThermodynamicsError(ThermodynamicsError.BAROSTATED_NONPERIODIC)            # states.py:1764-1766
        self._pressure = None if value is None else float(to_md(value))

    @property
    def barostat(self):
        """states.py:705-727: what the reference returns as a copy of the System's MonteCarloBarostat -- here the three numbers it
        carries (pressure of this state, its temperature, the attempt frequency); None at constant volume."""
        if self._pressure is None:
            return None
        return MonteCarloBarostatSettings(self._pressure, self._temperature, self.barostat_frequency)

    @property
    def surface_tension(self):
        """states.py:729-748: None -- a surface tension needs a System with a membrane barostat, which this package does not build."""
        return None

    def get_volume(self, ignore_ensemble=False):
        """states.py:773-794: volume of the System's default periodic box (nm^3); None when the volume fluctuates (a pressure is set,
        unless ignore_ensemble) or the System is not periodic."""
        if self._pressure is not None and not ignore_ensemble:
            return None
        if not self._system.usesPeriodicBoundaryConditions():
            return None
        return float(abs(np.linalg.det(np.array(self._system.getDefaultPeriodicBoxVectors(), dtype=np.float64).reshape(3, 3))))

    @property
    def volume(self):
        """states.py:763-771."""
        return self.get_volume()

    @property
    def system(self):
        return self._system

    def get_system(self, remove_thermostat=False, remove_barostat=False):
        """states.py:836-870.  A System here holds neither a thermostat nor a barostat force (temperature and pressure live on the
        state and go to the engine), so there is nothing to remove: the two flags are accepted for the caller's sake."""
        return self._system

    @property
    def temperature(self):
        return self._temperature

    @temperature.setter
    def temperature(self, value):
        if value is None:
            raise ThermodynamicsError(ThermodynamicsError.NONE_TEMPERATURE)                  # states.py:664-666
        value = float(to_md(value))                    # float kelvin or an openmm.unit.Quantity
        if not value > 0:
            raise ValueError('temperature must be positive')
        self._temperature = value

    @property
    def kT(self):
        return constants.kB * self._temperature

    @property
    def beta(self):
        return 1.0 / (constants.kB * self._temperature)

    @property
    def n_particles(self):
        return self._system.getNumParticles()

    @property
    def is_periodic(self):
        return self._system.usesPeriodicBoundaryConditions()

    @property
    def default_box_vectors(self):
        return self._system.getDefaultPeriodicBoxVectors()

    # lambda parameters: a plain ThermodynamicState is the fully interacting end state
    lambda_sterics = 1.0
    lambda_electrostatics = 1.0

    def region_lambdas(self, names):
        return [1.0] * len(names), [1.0] * len(names)

    def region_bonded_lambdas(self, names):
        return [1.0] * len(names), [1.0] * len(names), [1.0] * len(names)

    def __setstate__(self, state):
        """States pickled before ``pressure`` / ``temperature`` became properties (storage format 'openmmtools_amd-records-1'
        of earlier revisions) carry the plain attribute names: map them onto the backing fields so that old stores resume."""
        state = dict(state)
        for old, new in (('pressure', '_pressure'), ('temperature', '_temperature'), ('system', '_system')):
            if old in state and new not in state:
                state[new] = state.pop(old)
        state.setdefault('_pressure', None)
        state.setdefault('barostat_frequency', 25)
        self.__dict__.update(state)

    @staticmethod
    def _compute_reduced_potential(potential_energy, temperature, volume=None, pressure=None):
        """states.py:1908-1917: u = beta (U + p V); energies per mole, so N_A is already folded in."""
        beta = 1.0 / (constants.kB * temperature)
        reduced = potential_energy
        if pressure is not None:
            reduced = reduced + pressure * volume
        return beta * reduced

    def reduced_potential(self, context_state):
        """states.py:932-992.  Reduced potential of a SamplerState with a cached potential energy, or of an energy in kJ/mol (the
        reference also takes an openmm.Context there, hence the argument's name)."""
        sampler_state_or_energy = context_state
        if isinstance(sampler_state_or_energy, SamplerState):
            energy = sampler_state_or_energy.potential_energy
            volume = sampler_state_or_energy.volume
            if energy is None:
                raise ValueError('SamplerState has no cached potential energy')
        else:
            energy, volume = float(sampler_state_or_energy), None
        return self._compute_reduced_potential(energy, self._temperature, volume, self.pressure)

    def is_state_compatible(self, thermodynamic_state):
        """states.py:994-1050: same standard system and same ensemble."""
        other = thermodynamic_state
        return (self._system is other._system or self._system.fingerprint() == other._system.fingerprint()) \
            and (self.pressure is None) == (other.pressure is None)

    def __deepcopy__(self, memo):
        # Systems are treated as immutable once wrapped (the reference deep-copies and re-hashes)
        new = copy.copy(self)
        return new


class AlchemicalStateError(ValueError):
    """alchemy.py:60-62 (a ValueError here as well: what this class raised before it had a name of its own)."""


class AlchemicalState:
    """lambda_sterics / lambda_electrostatics carrier (alchemy.py:86-410, only these two).  ``parameters_name_suffix`` (states.py:3149-3181,
    alchemy.py:203-231): the state of the alchemical region of that name -- its parameters are then also reachable as
    ``lambda_sterics_<suffix>`` / ``lambda_electrostatics_<suffix>``, the names the factory gives the region's global parameters
    (alchemy.py:1360-1377)."""

    _PARAMETERS = ('lambda_sterics', 'lambda_electrostatics', 'lambda_bonds', 'lambda_angles', 'lambda_torsions')

    def __init__(self, lambda_sterics=1.0, lambda_electrostatics=1.0, lambda_bonds=1.0, lambda_angles=1.0, lambda_torsions=1.0,
                 parameters_name_suffix=None, **kwargs):
        object.__setattr__(self, 'parameters_name_suffix', parameters_name_suffix)
        values = dict(lambda_sterics=lambda_sterics, lambda_electrostatics=lambda_electrostatics, lambda_bonds=lambda_bonds,
                      lambda_angles=lambda_angles, lambda_torsions=lambda_torsions)
        for key, value in kwargs.items():                       # lambda_sterics_<suffix>=... as the reference's constructor takes them
            base = self._base_name(key)
            if base not in self._PARAMETERS:
                raise AlchemicalStateError('Unknown parameters %r' % key)                    # states.py:3170-3172
            values[base] = value
        # lambda_bonds / angles / torsions act on the bonded terms a region names (AlchemicalRegion.alchemical_bonds ..., alchemy.py:196-199);
        # None = "the System does not define it" (alchemy.py:94-99) is the interacting value here
        for name, value in values.items():
            setattr(self, name, 1.0 if value is None else float(value))
        # the bonded parameters this state was GIVEN (the reference keeps the others at None = "not defined", alchemy.py:94-99, and
        # set_alchemical_parameters leaves those alone, :247-262)
        given = {self._base_name(k) for k in kwargs}
        object.__setattr__(self, '_defined', {k for k in ('lambda_bonds', 'lambda_angles', 'lambda_torsions')
                                              if k in given or (values[k] is not None and float(values[k]) != 1.0)})

    parameters_name_suffix = None
    lambda_bonds = lambda_angles = lambda_torsions = 1.0          # (states pickled before these were carried)
    _defined = frozenset()

    def _base_name(self, name):
        sfx = self.parameters_name_suffix
        if sfx is not None and name.endswith('_' + sfx):
            return name[:-len(sfx) - 1]
        return name

    def __getattr__(self, name):                      # only reached for names that are not plain attributes
        base = self._base_name(name) if not name.startswith('__') else name
        if base != name and base in self._PARAMETERS:
            return self.__dict__[base]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        base = self._base_name(name)
        object.__setattr__(self, base if base in self._PARAMETERS else name, float(value) if base in self._PARAMETERS else value)

    def set_alchemical_parameters(self, new_value):
        """alchemy.py:247-262: every alchemical parameter this state controls to ``new_value``."""
        v = float(new_value)
        if not 0.0 <= v <= 1.0:
            raise ValueError('{} must be between 0 and 1.'.format('lambda_sterics'))     # alchemy.py:216-218: the first parameter's validator speaks
        self.lambda_sterics = v
        self.lambda_electrostatics = v
        for k in sorted(self.__dict__.get('_defined', ())):
            setattr(self, k, v)

    @staticmethod
    def _has_region(system, suffix):
        regions = getattr(system, 'alchemical_regions', None)
        if regions is not None:
            return any(r.name == suffix for r in regions)
        region = getattr(system, 'alchemical_region', None)
        return region is not None and (region.name == suffix or suffix is None or region.name is None)

    @classmethod
    def from_system(cls, system, *args, parameters_name_suffix=None, **kwargs):
        """alchemy.py:203-231: the state a System's alchemical parameters stand at (those written by apply_to_system, else the
        interacting end state the factory creates); a System without an alchemical region (of that name) has none."""
        if not cls._has_region(system, parameters_name_suffix):
            raise AlchemicalStateError('system has no alchemical region' + ('' if parameters_name_suffix is None else ' named %r' % parameters_name_suffix))
        stored = (getattr(system, 'alchemical_parameters', None) or {}).get(parameters_name_suffix, {})
        state = cls(*args, parameters_name_suffix=parameters_name_suffix, **dict({k: stored.get(k, 1.0) for k in cls._PARAMETERS}, **kwargs))
        # lambda_bonds / angles / torsions are defined where the region names softened terms (alchemy.py:1170-1197, 1252-1275, 1331-1354)
        regions = getattr(system, 'alchemical_regions', None) or [getattr(system, 'alchemical_region', None)]
        for r in regions:
            if r is not None and (r.name == parameters_name_suffix or len(regions) == 1):
                state._defined.update('lambda_' + k for k in ('bonds', 'angles', 'torsions') if getattr(r, 'alchemical_' + k, None))
        return state

    def apply_to_system(self, system):
        """alchemy.py:354-373: the System's own alchemical parameters (what a state read back with from_system starts from) set to
        this state's.  The engine takes lambdas from the thermodynamic states, not from the System."""
        if not self._has_region(system, self.parameters_name_suffix):
            raise AlchemicalStateError('system has no alchemical region')
        stored = dict(getattr(system, 'alchemical_parameters', None) or {})
        stored[self.parameters_name_suffix] = {k: float(getattr(self, k)) for k in self._PARAMETERS}
        system.alchemical_parameters = stored

    def check_system_consistency(self, system):
        """alchemy.py:375-393: AlchemicalStateError unless the System's alchemical parameters equal this state's."""
        other = type(self).from_system(system, parameters_name_suffix=self.parameters_name_suffix)
        for k in self._PARAMETERS:
            if float(getattr(other, k)) != float(getattr(self, k)):
                raise AlchemicalStateError('system has %s = %r, the state %r' % (k, getattr(other, k), getattr(self, k)))


class CompoundThermodynamicState(ThermodynamicState):
    """states.py:2524-3046 restricted to AlchemicalState composable states: one (any suffix), or one per named alchemical region."""

    def __init__(self, thermodynamic_state, composable_states):
        super().__init__(thermodynamic_state.system, thermodynamic_state.temperature, thermodynamic_state.pressure)
        if len(composable_states) < 1 or not all(isinstance(c, AlchemicalState) for c in composable_states):
            raise NotImplementedError('only AlchemicalState composable states are supported')
        suffixes = [c.parameters_name_suffix for c in composable_states]
        if len(set(suffixes)) != len(suffixes):
            raise ValueError('composable states control the same parameters')                 # states.py:2574-2590
        object.__setattr__(self, '_alchs', [copy.copy(c) for c in composable_states])

    @property
    def _alch(self):
        return self._alchs[0]

    def _owner(self, name):
        """the composable state that answers to the attribute ``name`` (plain or suffixed), or None"""
        for c in self.__dict__.get('_alchs', ()):
            sfx = c.parameters_name_suffix
            if name in AlchemicalState._PARAMETERS and (sfx is None or len(self._alchs) == 1):
                return c, name
            if sfx is not None and name.endswith('_' + sfx) and name[:-len(sfx) - 1] in AlchemicalState._PARAMETERS:
                return c, name[:-len(sfx) - 1]
        return None

    # (the plain names shadow the end-state class attributes of ThermodynamicState)
    @property
    def lambda_sterics(self):
        hit = self._owner('lambda_sterics')
        return 1.0 if hit is None else getattr(hit[0], hit[1])

    @property
    def lambda_electrostatics(self):
        hit = self._owner('lambda_electrostatics')
        return 1.0 if hit is None else getattr(hit[0], hit[1])

    def __getattr__(self, name):
        if not name.startswith('_'):
            hit = self._owner(name)
            if hit is not None:
                return getattr(hit[0], hit[1])
        raise AttributeError(name)

    def __setattr__(self, name, value):
        hit = None if name.startswith('_') else self._owner(name)
        if hit is not None:
            setattr(hit[0], hit[1], float(value))
        else:
            object.__setattr__(self, name, value)

    def _state_of_region(self, nm, n_names):
        c = next((c for c in self._alchs if c.parameters_name_suffix == nm), None)
        if c is None and len(self._alchs) == 1 and n_names == 1:
            c = self._alchs[0]
        return c

    def region_lambdas(self, names):
        """(lambda_sterics, lambda_electrostatics) of the alchemical regions ``names`` at this state: the composable state with that
        suffix (a single unsuffixed state answers for a single region)."""
        cs = [self._state_of_region(nm, len(names)) for nm in names]
        return [1.0 if c is None else c.lambda_sterics for c in cs], [1.0 if c is None else c.lambda_electrostatics for c in cs]

    def region_bonded_lambdas(self, names):
        """(lambda_bonds, lambda_angles, lambda_torsions) of the regions ``names``"""
        cs = [self._state_of_region(nm, len(names)) for nm in names]
        return tuple([1.0 if c is None else getattr(c, k) for c in cs] for k in ('lambda_bonds', 'lambda_angles', 'lambda_torsions'))

    def __deepcopy__(self, memo):
        new = copy.copy(self)
        object.__setattr__(new, '_alchs', [copy.copy(c) for c in self._alchs])
        return new

    def __setstate__(self, state):
        state = dict(state)
        if '_alch' in state and '_alchs' not in state:           # stores written before several composable states were carried
            state['_alchs'] = [state.pop('_alch')]
        super().__setstate__(state)


class SamplerState:
    def __init__(self, positions, velocities=None, box_vectors=None):
        # plain arrays in nm / nm ps^-1, or openmm.unit.Quantity of arrays / Vec3 lists (states.py:1975-2010)
        self.positions = np.array(to_md(positions), dtype=np.float64).reshape(-1, 3)
        self.velocities = None if velocities is None else np.array(to_md(velocities), dtype=np.float64).reshape(-1, 3)
        if self.velocities is not None and self.velocities.shape != self.positions.shape:
            raise SamplerStateError(SamplerStateError.INCONSISTENT_VELOCITIES)               # states.py:2397-2399
        if box_vectors is not None and not hasattr(box_vectors, 'value_in_unit_system'):
            box_vectors = [to_md(b) for b in box_vectors] if isinstance(box_vectors, (list, tuple)) else box_vectors
        self.box_vectors = None if box_vectors is None else np.array(to_md(box_vectors), dtype=np.float64).reshape(3, 3)
        self.potential_energy = None
        self.kinetic_energy = None

    @property
    def n_particles(self):
        return self.positions.shape[0]

    @property
    def total_energy(self):
        """states.py:2170-2174: potential + kinetic energy when both are cached, else None."""
        if self.potential_energy is None or self.kinetic_energy is None:
            return None
        return self.potential_energy + self.kinetic_energy

    @property
    def area_xy(self):
        """states.py:2180-2184: area of the box in the xy plane (None without a box)."""
        if self.box_vectors is None:
            return None
        return float(abs(np.cross(self.box_vectors[0], self.box_vectors[1])[2]))

    def __getitem__(self, item):
        """states.py:2296-2325: the sampler state of a subset of particles (an index, a slice or a sequence of indices): copies
        of their positions and velocities, the same box, no energies (undefined for a subset)."""
        if np.issubdtype(type(item), np.integer):
            item = [int(item)]
        sub = SamplerState(np.array(self.positions[item], dtype=np.float64),
                           velocities=None if self.velocities is None else np.array(self.velocities[item], dtype=np.float64),
                           box_vectors=None if self.box_vectors is None else self.box_vectors.copy())
        return sub

    def has_nan(self):
        """states.py:2281-2293: any NaN among the positions (what the reference checks) or the velocities."""
        return bool(np.isnan(self.positions).any() or (self.velocities is not None and np.isnan(self.velocities).any()))

    @property
    def volume(self):
        if self.box_vectors is None:
            return None
        return float(abs(np.linalg.det(self.box_vectors)))

    @property
    def box_edges(self):
        """Orthorhombic edge lengths; triclinic boxes are not supported by the engine."""
        if self.box_vectors is None:
            return None
        off = self.box_vectors - np.diag(np.diag(self.box_vectors))
        if np.abs(off).max() > 1e-9:
            raise NotImplementedError('triclinic periodic boxes are not supported')
        return np.diag(self.box_vectors).copy()

    def __getstate__(self):
        return dict(positions=self.positions, velocities=self.velocities, box_vectors=self.box_vectors,
                    potential_energy=self.potential_energy, kinetic_energy=self.kinetic_energy)

    def __setstate__(self, s):
        self.__dict__.update(s)


def reduced_potential_at_states(sampler_state, thermodynamic_states, context_cache=None, engine=None):
    """states.py:144-183: the reduced potentials of ONE configuration at a list of thermodynamic states (grouped by compatible
    System: one set of device tables per group, every state of a group from one evaluation where only the temperature or
    lambda differs).  The engine's u_kl row for a single replica: ``engine`` is the engine object to use (default: a HipEngine
    on the GPU; ``context_cache.make_engine()`` when a cache is given)."""
    from .multistate import MultiStateSampler
    from . import mcmc
    if engine is None and context_cache is not None and hasattr(context_cache, 'make_engine'):
        engine = context_cache.make_engine()
    row = MultiStateSampler(mcmc_moves=mcmc.LangevinDynamicsMove(n_steps=1), number_of_iterations=0, engine=engine,
                            online_analysis_interval=None)
    row.create(list(thermodynamic_states), [sampler_state], storage=None)
    row._compute_energies()
    return np.array(row.energy_thermodynamic_states[0], dtype=np.float64)


def group_by_compatibility(states):
    """states.py:186-217."""
    groups, indices = [], []
    for i, s in enumerate(states):
        for g, idx in zip(groups, indices):
            if s.is_state_compatible(g[0]):
                g.append(s)
                idx.append(i)
                break
        else:
            groups.append([s])
            indices.append([i])
    return groups, indices
