"""Langevin MCMC moves as the sampler's propagation recipe.

Mirrors openmmtools/mcmc.py: SequenceMove (:350-440), BaseIntegratorMove (:603-807), LangevinDynamicsMove (:1066-1172),
LangevinSplittingDynamicsMove (:1180-1316), GHMCMove (:1323-1490), HMCMove (:1493-1590), MonteCarloBarostatMove (:1597-1700), MetropolizedMove (:810-975),
MCDisplacementMove (:1704-1770), MCRotationMove (:1777-1910).  In the reference ``apply`` pushes one replica
through an OpenMM Context (:668-776); here a move only carries the parameters and the
multistate sampler propagates *all* replicas in one batched device call
(_engine.HipEngine.propagate -> remd_propagate).
"""
import numpy as np
from . import unit, integrators


class MCMCMove:
    """mcmc.py:143-213.  A move carries parameters (and statistics); the multistate samplers apply it to all their replicas in one
    batched engine call.  ``apply`` keeps the reference's single-configuration entry point (:155-170): the configuration goes
    through the same engine as a one-replica ensemble and ``sampler_state`` is updated in place."""

    def apply(self, thermodynamic_state, sampler_state, context_cache=None, engine=None):
        """``context_cache`` is accepted for call compatibility (a cache with ``make_engine`` chooses the device);
        ``engine``: the engine object to run on (default: a HipEngine on the GPU -- there is no CPU fallback)."""
        import numpy as np
        if engine is None and context_cache is not None and hasattr(context_cache, 'make_engine'):
            engine = context_cache.make_engine()
        # the reference rebuilds the integrator and re-applies the state on every apply (mcmc.py:692-700): a state or a move
        # mutated between two calls (temperature, pressure, lambda, n_steps, timestep ...) must not meet a stale driver, so
        # their parameters are part of the key and a mismatch builds a new one
        key = (id(thermodynamic_state), id(engine) if engine is not None else None, sampler_state.n_particles,
               _parameter_fingerprint(thermodynamic_state), _parameter_fingerprint(self))
        held = self.__dict__.get('_apply_driver')
        if held is None or held[0] != key:
            driver = MCMCSampler(thermodynamic_state, sampler_state, self, engine=engine)
            driver._ensemble()
            self.__dict__['_apply_driver'] = held = (key, driver, thermodynamic_state)        # (keeps the state alive: ids stay unique)
        else:
            driver = held[1]
            d = driver._ensemble()
            box = sampler_state.box_edges if thermodynamic_state.is_periodic else np.zeros(3)
            d._engine.set_replicas(1, 0, sampler_state.positions[None], None if sampler_state.velocities is None else sampler_state.velocities[None],
                                   np.asarray(box, dtype=np.float64)[None], d._replica_thermodynamic_states)
            d._sampler_states_stale = True
        driver.run(1)
        new = driver.sampler_state
        sampler_state.potential_energy, sampler_state.kinetic_energy = new.potential_energy, new.kinetic_energy
        sampler_state.positions = new.positions.copy()
        sampler_state.velocities = None if new.velocities is None else new.velocities.copy()
        if new.box_vectors is not None:
            sampler_state.box_vectors = new.box_vectors.copy()

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('_apply_driver', None)            # the engine behind ``apply`` is not part of the move
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)


def _parameter_fingerprint(obj, _depth=0):
    """Hashable digest of the parameters an engine was programmed with (numbers, strings, flags; nested moves / composable
    states followed; systems by their content hash; statistics and caches skipped)."""
    if _depth > 4:
        return None
    if isinstance(obj, (int, float, str, bool, type(None))):
        return obj
    if isinstance(obj, np.ndarray):
        return ('nd', obj.shape, obj.tobytes() if obj.size <= 64 else hash(obj.tobytes()))
    if isinstance(obj, (list, tuple)):
        return tuple(_parameter_fingerprint(o, _depth + 1) for o in obj)
    if hasattr(obj, 'fingerprint') and callable(obj.fingerprint):            # System: by CONTENT (one mutated in place must not meet
        return ('system', obj.fingerprint())                                  # the driver programmed with its old parameters)
    d = getattr(obj, '__dict__', None)
    if d is None:
        return repr(obj)
    skip = ('_apply_driver', 'statistics', 'n_accepted', 'n_proposed', 'n_attempted', '_standard_system_hash')
    return (type(obj).__name__,) + tuple((k, _parameter_fingerprint(v, _depth + 1)) for k, v in sorted(d.items())
                                         if k not in skip and not k.startswith('_cache'))


class MCMCSampler:
    """mcmc.py:216-347: one thermodynamic state, one configuration, one move applied ``n_iterations`` times.  The reference calls
    ``move.apply`` in a loop; here the same engine that propagates the replicas of a multistate sampler runs it as a
    one-replica, one-state ensemble (no mixing, nothing stored), so every move the engine knows is available unchanged."""

    def __init__(self, thermodynamic_state, sampler_state, move, engine=None, seed=None):
        import copy
        if seed is None:              # independent runs draw independent noise, like the reference's global streams
            seed = int(np.random.randint(0, 2 ** 31 - 1)) << 16 | int(np.random.randint(0, 2 ** 16))
        self.thermodynamic_state = copy.deepcopy(thermodynamic_state)       # :247-249
        self.sampler_state = copy.deepcopy(sampler_state)
        self.move = move
        self._engine, self._seed, self._driver = engine, seed, None

    def _ensemble(self):
        if self._driver is None:
            from .multistate import MultiStateSampler
            d = MultiStateSampler(mcmc_moves=[self.move], number_of_iterations=0, engine=self._engine, seed=self._seed,
                                  online_analysis_interval=None)
            d.create([self.thermodynamic_state], [self.sampler_state], storage=None)
            d._mcmc_moves = [self.move]                                      # the caller's object collects the statistics
            d._program_engine_move()
            self._driver = d
        return self._driver

    def run(self, n_iterations=1, context_cache=None):
        """:251-266 (context_cache: the reference's per-call cache of OpenMM Contexts; the engine of this sampler is its counterpart and
        is chosen at construction -- accepted and not used)."""
        if isinstance(self.move, WeightedMove):                              # a choice per application: the move's own apply
            for _ in range(int(n_iterations)):
                self.move.apply(self.thermodynamic_state, self.sampler_state, engine=self._engine)
            return
        d = self._ensemble()
        d.extend(int(n_iterations))
        self.sampler_state = d.sampler_states[0]
        # states.py:2431-2490 update_from_context: the sampler state leaves a move with its energies (kJ/mol)
        _, _, u, k = d._engine.get_replicas(positions=False, velocities=False, potential=True, kinetic=True)
        self.sampler_state.potential_energy = float(np.asarray(u).reshape(-1)[0])
        self.sampler_state.kinetic_energy = float(np.asarray(k).reshape(-1)[0])

    def minimize(self, tolerance=1.0 * unit.kilocalories_per_mole / unit.angstroms, max_iterations=100, context_cache=None):
        """:268-300 (the engine's FIRE minimiser, as MultiStateSampler.minimize; context_cache as in run)."""
        d = self._ensemble()
        d.minimize(tolerance=tolerance, max_iterations=max_iterations)
        self.sampler_state = d.sampler_states[0]


class SequenceMove(MCMCMove):
    """mcmc.py:350-440: the moves are applied in order, once per iteration."""

    def __init__(self, move_list, **kwargs):
        self.move_list = list(move_list)

    @property
    def statistics(self):
        return [getattr(m, 'statistics', None) for m in self.move_list]

    @statistics.setter
    def statistics(self, value):
        for m, v in zip(self.move_list, value):
            if hasattr(m, 'statistics'):
                m.statistics = v

    def __iter__(self):
        return iter(self.move_list)

    def __len__(self):
        return len(self.move_list)


class WeightedMove(MCMCMove):
    """mcmc.py:439-535: one move of a set per application, picked with probability equal to its weight (numpy's global stream,
    as in the reference).  The choice is made per configuration, so it runs through ``apply`` / ``MCMCSampler``; a multistate
    sampler propagates all replicas with ONE program per launch and refuses it (``SequenceMove`` is the batched composition)."""

    def __init__(self, move_set, **kwargs):
        self.move_set = list(move_set)

    @property
    def statistics(self):
        return [getattr(move, 'statistics', {}) for move, _ in self.move_set]

    @statistics.setter
    def statistics(self, value):
        for (move, _), v in zip(self.move_set, value):
            if hasattr(move, 'statistics'):
                move.statistics = v

    def apply(self, thermodynamic_state, sampler_state, context_cache=None, engine=None):
        import numpy as np
        moves, weights = zip(*self.move_set)
        move = moves[int(np.random.choice(len(moves), p=np.asarray(weights, dtype=np.float64)))]     # :516-517
        move.apply(thermodynamic_state, sampler_state, context_cache=context_cache, engine=engine)

    def __iter__(self):
        return iter(self.move_set)

    def __len__(self):
        return len(self.move_set)

    def __str__(self):
        return str(self.move_set)


class IntegratorMoveError(Exception):
    """mcmc.py:538-600: raised when a NaN is found after applying a move (after ``n_restart_attempts`` retries), carrying what
    is needed to reproduce it.  ``context``: dict(system=System, thermodynamic_state=..., positions, velocities, box) --
    the stand-in for the openmm.Context the reference keeps."""

    def __init__(self, message, move, context=None):
        super().__init__(message)
        self.move = move
        self.context = context

    def serialize_error(self, path_files_prefix):
        """mcmc.py:556-600: ``<prefix>-move.json`` (the move's parameters), ``<prefix>-system.xml`` (the serialised System, OpenMM
        document layout), ``<prefix>-integrator.json`` (the Langevin program the engine ran) and ``<prefix>-state.npz``
        (positions, velocities, box of the failed replica).  Existing files are overwritten."""
        import json
        import os
        import numpy as np
        directory_path = os.path.dirname(path_files_prefix)
        if directory_path and not os.path.exists(directory_path):
            os.makedirs(directory_path)
        mv = {k: (v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)) for k, v in vars(self.move).items()}
        mv['class'] = type(self.move).__name__
        with open(path_files_prefix + '-move.json', 'w') as f:
            json.dump(mv, f)
        ctx = self.context or {}
        if ctx.get('system') is not None:
            try:
                from . import system_xml
                with open(path_files_prefix + '-system.xml', 'w') as f:
                    f.write(system_xml.to_xml(ctx["system"], pressure=getattr(ctx.get("thermodynamic_state"), "pressure", None), temperature=getattr(ctx.get("thermodynamic_state"), "temperature", None)))
            except Exception as exc:                       # a System the XML writer does not cover: say so instead of failing the dump
                with open(path_files_prefix + '-system.xml', 'w') as f:
                    f.write('<!-- System not serialisable: %s -->\n' % exc)
        ts = ctx.get('thermodynamic_state')
        integ = dict(kind='LangevinIntegrator', splitting=getattr(self.move, 'splitting', None), timestep_ps=getattr(self.move, 'timestep', None),
                     collision_rate_per_ps=getattr(self.move, 'collision_rate', None), n_steps=getattr(self.move, 'n_steps', None),
                     temperature_K=getattr(ts, 'temperature', None), pressure=getattr(ts, 'pressure', None),
                     lambda_sterics=getattr(ts, 'lambda_sterics', None), lambda_electrostatics=getattr(ts, 'lambda_electrostatics', None))
        with open(path_files_prefix + '-integrator.json', 'w') as f:
            json.dump(integ, f)
        arrays = {k: np.asarray(ctx[k]) for k in ('positions', 'velocities', 'box', 'positions_before', 'velocities_before') if ctx.get(k) is not None}
        np.savez(path_files_prefix + '-state.npz', **arrays)


class BaseIntegratorMove(MCMCMove):
    def __init__(self, n_steps, reassign_velocities=False, n_restart_attempts=4):
        self.n_steps = int(n_steps)
        self.reassign_velocities = bool(reassign_velocities)
        self.n_restart_attempts = int(n_restart_attempts)
        self.statistics = {}


class LangevinSplittingDynamicsMove(BaseIntegratorMove):
    def __init__(self, timestep=1.0 * unit.femtosecond, collision_rate=10.0 / unit.picoseconds,
                 n_steps=1000, reassign_velocities=False, splitting="V R O R V", constraint_tolerance=1.0e-8,
                 measure_shadow_work=False, measure_heat=False, **kwargs):
        super().__init__(n_steps=n_steps, reassign_velocities=reassign_velocities, **kwargs)
        self.timestep = float(unit.to_md(timestep))                 # ps; float or openmm.unit.Quantity
        self.collision_rate = float(unit.to_md(collision_rate))     # 1/ps
        self.splitting = splitting
        self.constraint_tolerance = float(constraint_tolerance)
        self.measure_shadow_work = bool(measure_shadow_work)         # mcmc.py:1290-1291: passed through to the integrator
        self.measure_heat = bool(measure_heat)

    def _get_integrator(self, thermodynamic_state):
        """mcmc.py:1308-1316."""
        return integrators.LangevinIntegrator(temperature=thermodynamic_state.temperature,
                                              collision_rate=self.collision_rate, timestep=self.timestep,
                                              splitting=self.splitting,
                                              constraint_tolerance=self.constraint_tolerance,
                                              measure_shadow_work=self.measure_shadow_work, measure_heat=self.measure_heat)


class LangevinDynamicsMove(LangevinSplittingDynamicsMove):
    """mcmc.py:1066-1172.  The reference uses openmm.LangevinMiddleIntegrator (:1169): per step a FULL kick, half a drift, the
    Ornstein-Uhlenbeck update, half a drift -- "V R O R" with velocities kept half a step behind (leapfrog).  Run here as "V R O R V":
    two adjacent half kicks see the same positions, so they merge into the middle scheme's full kick -- the positions of the two programs
    are IDENTICAL step by step for the same noise, and the velocities differ by the half kick v(t) = v_leapfrog + dt F(x_t) / 2m, i.e.
    the engine holds the on-step velocity (the one OpenMM's kinetic energy of a leapfrog integrator is evaluated at).
    tests/test_mc_moves.py::test_langevin_dynamics_move_is_the_leapfrog_middle_scheme holds both statements at 1e-15."""

    def __init__(self, timestep=1.0 * unit.femtosecond, collision_rate=10.0 / unit.picoseconds,
                 n_steps=1000, reassign_velocities=False, constraint_tolerance=1e-8, **kwargs):
        super().__init__(timestep=timestep, collision_rate=collision_rate, n_steps=n_steps,
                         reassign_velocities=reassign_velocities, splitting="V R O R V",
                         constraint_tolerance=constraint_tolerance, **kwargs)


class IntegratorMove(LangevinSplittingDynamicsMove):
    """mcmc.py:977-1020: ``n_steps`` of a caller-supplied integrator.  The engine runs Langevin splittings, so the integrator
    must be one of ``openmmtools_amd.integrators`` (LangevinIntegrator and its named splittings); its temperature is replaced
    by the thermodynamic state's when the move is applied, as the reference's thermostated integrators are (:676-680)."""

    def __init__(self, integrator, n_steps, **kwargs):
        if not isinstance(integrator, integrators.LangevinIntegrator):
            raise NotImplementedError('IntegratorMove runs the Langevin-splitting integrators of openmmtools_amd.integrators; got %s'
                                      % type(integrator).__name__)
        super().__init__(timestep=integrator.getStepSize(), collision_rate=integrator._gamma, n_steps=n_steps,
                         splitting=integrator.splitting, constraint_tolerance=integrator.getConstraintTolerance(),
                         measure_shadow_work=integrator._measure_shadow_work and not integrator.is_metropolized,
                         measure_heat=integrator.measure_heat, **kwargs)
        self.integrator = integrator

    def _get_integrator(self, thermodynamic_state):
        import copy
        integ = copy.deepcopy(self.integrator)                 # :1003-1009: a copy per application
        integ.setTemperature(thermodynamic_state.temperature)
        return integ


class GHMCMove(LangevinSplittingDynamicsMove):
    """mcmc.py:1323-1490: generalized hybrid Monte Carlo -- ``n_steps`` of GHMCIntegrator, i.e. the Metropolized splitting
    "O { V R V } O" (integrators.py:2286), whose accepted / attempted steps the move accumulates (the reference reads the
    integrator's ``naccept`` / ``ntrials`` after the integration, :1478-1489; here the engine's per-replica counters,
    remd_get_work, are credited to the move of the state a replica was propagated in)."""

    def __init__(self, timestep=1.0 * unit.femtosecond, collision_rate=20.0 / unit.picoseconds, n_steps=1000, **kwargs):
        super().__init__(timestep=timestep, collision_rate=collision_rate, n_steps=n_steps,
                         splitting=integrators.GHMCIntegrator.SPLITTING, **kwargs)
        self.n_accepted = 0      # :1399-1400
        self.n_proposed = 0

    @property
    def fraction_accepted(self):
        """:1403-1412: accepted over attempted steps, NaN before the first one."""
        if self.n_proposed == 0:
            return float('nan')
        return float(self.n_accepted) / self.n_proposed

    @property
    def statistics(self):
        return dict(n_accepted=self.n_accepted, n_proposed=self.n_proposed)

    @statistics.setter
    def statistics(self, value):
        self.n_accepted = int(value.get('n_accepted', 0))
        self.n_proposed = int(value.get('n_proposed', 0))

    def reset_statistics(self):
        self.n_accepted = 0
        self.n_proposed = 0

    def _get_integrator(self, thermodynamic_state):
        """:1469-1474."""
        return integrators.GHMCIntegrator(temperature=thermodynamic_state.temperature, collision_rate=self.collision_rate,
                                          timestep=self.timestep, constraint_tolerance=self.constraint_tolerance)


class HMCMove(LangevinSplittingDynamicsMove):
    """mcmc.py:1493-1590: hybrid Monte Carlo -- velocities from the Maxwell-Boltzmann distribution, ``n_steps`` velocity Verlet
    steps, one Metropolis test on the change of total energy (HMCIntegrator, integrators.py:885-1010).  As a program of the
    Langevin engine that is ONE step of the splitting

        O { V R V  V R V  ...  V R V }            (n_steps groups)

    at timestep n_steps * dt with an infinite collision rate: O then draws v = sqrt(kT/m) xi, every V is the half kick dt/2
    (2 n_steps of them share the step), every R the drift dt, and the braces accept or restore the start of the trajectory.
    Like the reference, ``apply`` runs the integrator ``n_steps`` times (mcmc.py:719 steps the HMCIntegrator(nsteps=n_steps)
    n_steps times): n_steps trajectories of n_steps steps per iteration."""

    RESAMPLE = 1.0e12          # 1/ps: exp(-RESAMPLE * dt) is exactly 0 for any dt the integrators accept

    def __init__(self, timestep=1.0 * unit.femtosecond, n_steps=1000, **kwargs):
        n = int(n_steps)
        if n < 1:
            raise ValueError('HMCMove needs at least one step per trajectory')
        super().__init__(timestep=timestep, collision_rate=self.RESAMPLE, n_steps=n,
                         splitting='O {' + ' V R V' * n + ' }', **kwargs)

    @property
    def engine_timestep(self):
        """ps per pass of the splitting (``timestep`` keeps the reference's meaning: the velocity Verlet step)."""
        return self.timestep * self.n_steps

    def _get_integrator(self, thermodynamic_state):
        return integrators.LangevinIntegrator(temperature=thermodynamic_state.temperature, collision_rate=self.collision_rate,
                                              timestep=self.engine_timestep, splitting=self.splitting,
                                              constraint_tolerance=self.constraint_tolerance)


class MonteCarloBarostatMove(BaseIntegratorMove):
    """mcmc.py:1597-1700: ``n_attempts`` Monte Carlo volume moves (the reference steps a DummyIntegrator ``n_attempts`` times
    with the state's MonteCarloBarostat at frequency 1).  The thermodynamic state must carry a pressure."""

    def __init__(self, n_attempts=5, **kwargs):
        super().__init__(n_steps=n_attempts, **kwargs)

    @property
    def n_attempts(self):
        return self.n_steps

    @n_attempts.setter
    def n_attempts(self, value):
        self.n_steps = int(value)


class MetropolizedMove(MCMCMove):
    """mcmc.py:810-975: propose new positions for a subset of atoms, accept with min(1, exp(-delta u)) on the reduced potential
    of the replica's own state, restore otherwise.  The reference pushes one replica through a Context twice; the sampler
    here does it for all local replicas at once (positions of every replica proposed on the host, two batched energy
    evaluations on the device; multistatesampler._apply_metropolized_move).  Subclasses implement ``_propose_positions``."""

    def __init__(self, atom_subset=None, **kwargs):
        self.n_accepted = 0
        self.n_proposed = 0
        self.atom_subset = atom_subset

    @property
    def statistics(self):
        return dict(n_accepted=self.n_accepted, n_proposed=self.n_proposed)

    @statistics.setter
    def statistics(self, value):
        self.n_accepted = int(value.get('n_accepted', 0))
        self.n_proposed = int(value.get('n_proposed', 0))

    def _subset(self):
        """:873-879: all atoms by default; a one-element list becomes a slice so that the proposal sees an (1, 3) array."""
        import numpy as np
        if self.atom_subset is None:
            return slice(None)
        if not isinstance(self.atom_subset, slice) and len(self.atom_subset) == 1:
            return slice(int(self.atom_subset[0]), int(self.atom_subset[0]) + 1)
        return self.atom_subset if isinstance(self.atom_subset, slice) else np.asarray(self.atom_subset, dtype=np.int64)

    def _propose_positions(self, positions, rng=None):
        raise NotImplementedError('MetropolizedMove subclasses propose the new positions')


def _uniform_source(rng):
    import numpy as np
    return np.random if rng is None else rng


class MCDisplacementMove(MetropolizedMove):
    """mcmc.py:1704-1770: rigid translation of the subset by a normal vector of standard deviation ``displacement_sigma``."""

    def __init__(self, displacement_sigma=1.0 * unit.nanometer, **kwargs):
        super().__init__(**kwargs)
        self.displacement_sigma = float(unit.to_md(displacement_sigma))      # nm

    @staticmethod
    def displace_positions(positions, displacement_sigma=1.0 * unit.nanometer, rng=None):
        """:1744-1765 (positions in nm; ``rng``: a numpy Generator, default the global numpy stream as in the reference)."""
        import numpy as np
        src = _uniform_source(rng)
        vector = (src.standard_normal(3) if rng is not None else src.randn(3)) * float(unit.to_md(displacement_sigma))
        return np.asarray(positions, dtype=np.float64) + vector

    def _propose_positions(self, positions, rng=None):
        return self.displace_positions(positions, self.displacement_sigma, rng)


class MCRotationMove(MetropolizedMove):
    """mcmc.py:1777-1910: uniform random rotation of the subset about its centre of geometry (Shoemake's quaternions)."""

    @classmethod
    def rotate_positions(cls, positions, rng=None):
        import numpy as np
        x = np.asarray(positions, dtype=np.float64)
        centre = x.mean(0)                                                   # :1822
        return (cls.generate_random_rotation_matrix(rng) @ (x - centre).T).T + centre

    @classmethod
    def generate_random_rotation_matrix(cls, rng=None):
        return cls._rotation_matrix_from_quaternion(cls._generate_uniform_quaternion(rng))

    @staticmethod
    def _rotation_matrix_from_quaternion(q):
        """:1842-1880: the quaternion need not be normalised (zero norm gives the identity)."""
        import numpy as np
        w, x, y, z = (float(c) for c in q)
        n = w * w + x * x + y * y + z * z
        s = 2.0 / n if n > 0.0 else 0.0
        return np.array([[1.0 - s * (y * y + z * z), s * (x * y - w * z), s * (x * z + w * y)],
                         [s * (x * y + w * z), 1.0 - s * (x * x + z * z), s * (y * z - w * x)],
                         [s * (x * z - w * y), s * (y * z + w * x), 1.0 - s * (x * x + y * y)]])

    @staticmethod
    def _generate_uniform_quaternion(rng=None):
        """:1882-1906."""
        import numpy as np
        src = _uniform_source(rng)
        u = src.random(3) if rng is not None else src.rand(3)
        return np.array([np.sqrt(1 - u[0]) * np.sin(2 * np.pi * u[1]), np.sqrt(1 - u[0]) * np.cos(2 * np.pi * u[1]),
                         np.sqrt(u[0]) * np.sin(2 * np.pi * u[2]), np.sqrt(u[0]) * np.cos(2 * np.pi * u[2])])

    def _propose_positions(self, positions, rng=None):
        return self.rotate_positions(positions, rng)
