"""Langevin MCMC moves as the sampler's propagation recipe.

Mirrors openmmtools/mcmc.py: SequenceMove (:350-440), BaseIntegratorMove (:603-807), LangevinDynamicsMove (:1066-1172),
LangevinSplittingDynamicsMove (:1180-1316), MonteCarloBarostatMove (:1597-1700).  In the reference ``apply`` pushes one replica
through an OpenMM Context (:668-776); here a move only carries the parameters and the
multistate sampler propagates *all* replicas in one batched device call
(_engine.HipEngine.propagate -> remd_propagate).
"""
from . import unit, integrators


class MCMCMove:
    pass


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


class IntegratorMoveError(Exception):
    """mcmc.py:538-600 (the NaN error raised after n_restart_attempts)."""


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
        if measure_shadow_work or measure_heat:
            raise NotImplementedError('heat / shadow-work accumulators are not implemented')

    def _get_integrator(self, thermodynamic_state):
        """mcmc.py:1308-1316."""
        return integrators.LangevinIntegrator(temperature=thermodynamic_state.temperature,
                                              collision_rate=self.collision_rate, timestep=self.timestep,
                                              splitting=self.splitting,
                                              constraint_tolerance=self.constraint_tolerance)


class LangevinDynamicsMove(LangevinSplittingDynamicsMove):
    """mcmc.py:1066-1172.  The reference uses openmm.LangevinMiddleIntegrator (:1169), i.e. the
    BAOAB-equivalent "V R O R V" leapfrog-middle scheme; here it is run as that splitting."""

    def __init__(self, timestep=1.0 * unit.femtosecond, collision_rate=10.0 / unit.picoseconds,
                 n_steps=1000, reassign_velocities=False, constraint_tolerance=1e-8, **kwargs):
        super().__init__(timestep=timestep, collision_rate=collision_rate, n_steps=n_steps,
                         reassign_velocities=reassign_velocities, splitting="V R O R V",
                         constraint_tolerance=constraint_tolerance, **kwargs)


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
