"""states.create_thermodynamic_state_protocol (states.py:39-141) and the small SamplerState helpers."""
import numpy as np
import pytest
from openmmtools_amd import testsystems, states, unit, alchemy


def test_protocol_over_temperature_and_pressure():
    lj = testsystems.LennardJonesFluid(nparticles=216)
    out = states.create_thermodynamic_state_protocol(lj.system, dict(temperature=[100.0 * unit.kelvin, 120.0, 150.0]),
                                                     constants=dict(pressure=1.0 * unit.bar))
    assert [s.temperature for s in out] == [100.0, 120.0, 150.0]
    assert all(abs(s.pressure - 1.0 * unit.bar) < 1e-15 for s in out) and out[0] is not out[1]
    with pytest.raises(ValueError, match='No protocol'):
        states.create_thermodynamic_state_protocol(lj.system, {})
    with pytest.raises(ValueError, match='different lengths'):
        states.create_thermodynamic_state_protocol(lj.system, dict(temperature=[1.0, 2.0], pressure=[1.0]))
    with pytest.raises(ValueError, match='both in constants and protocol'):
        states.create_thermodynamic_state_protocol(lj.system, dict(temperature=[1.0]), constants=dict(temperature=2.0))
    with pytest.raises(ValueError, match='must specify the temperature'):
        states.create_thermodynamic_state_protocol(lj.system, dict(pressure=[1.0]))
    with pytest.raises(AttributeError, match='does not have protocol attribute'):
        states.create_thermodynamic_state_protocol(lj.system, dict(temperature=[100.0], lambda_bonds=[1.0]))


def test_protocol_over_alchemical_parameters():
    lj = testsystems.LennardJonesFluid(nparticles=216)
    asys = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, alchemy.AlchemicalRegion(alchemical_atoms=range(4)))
    base = states.ThermodynamicState(asys, 120.0 * unit.kelvin)
    out = states.create_thermodynamic_state_protocol(base, dict(lambda_sterics=[1.0, 0.5, 0.0], lambda_electrostatics=[1.0, 1.0, 1.0]),
                                                     composable_states=states.AlchemicalState.from_system(asys))
    assert [s.lambda_sterics for s in out] == [1.0, 0.5, 0.0] and all(isinstance(s, states.CompoundThermodynamicState) for s in out)
    assert all(s.temperature == 120.0 for s in out)
    out[0].lambda_sterics = 0.25
    assert out[1].lambda_sterics == 0.5                         # independent copies


def test_sampler_state_helpers_and_pressure_rules():
    ho = testsystems.HarmonicOscillator()
    ss = states.SamplerState(np.zeros((1, 3)), velocities=np.ones((1, 3)))
    assert ss.total_energy is None and not ss.has_nan()
    ss.potential_energy, ss.kinetic_energy = 2.0, 3.5
    assert ss.total_energy == 5.5
    ss.velocities[0, 1] = np.nan
    assert ss.has_nan()
    lj = testsystems.LennardJonesFluid(nparticles=216)
    st = states.ThermodynamicState(lj.system, 100.0)
    st.pressure = 2.0 * unit.bar
    assert abs(st.pressure - 2.0 * unit.bar) < 1e-15
    st.pressure = None
    assert st.pressure is None
    assert issubclass(states.ThermodynamicsError, Exception) and issubclass(states.SamplerStateError, Exception)


def test_context_cache_stand_in_keeps_the_configuration_surface():
    """openmmtools/cache.py:200-560: capacity / time_to_live / platform are kept; DeviceIndex names the engine's GPU."""
    import pickle
    from openmmtools_amd import cache
    assert len(cache.global_context_cache) == 0 and cache.global_context_cache.capacity is None
    cache.global_context_cache.platform = 'HIP'
    assert cache.global_context_cache.platform == 'HIP'
    cc = cache.ContextCache(capacity=None, time_to_live=None, platform='HIP', platform_properties={'DeviceIndex': '3'})
    assert cc.device_index == 3 and cache.ContextCache().device_index == 0
    with pytest.raises(ValueError, match='you need to also specify the platform'):
        cache.ContextCache(platform_properties={'DeviceIndex': '1'})
    back = pickle.loads(pickle.dumps(cc))
    assert back.device_index == 3 and back.capacity is None
    assert cache.DummyContextCache().capacity == 0
    cc.empty()
    from openmmtools_amd.multistate import MultiStateSampler
    sampler = MultiStateSampler()                              # tests/test_sampling.py:2328-2334
    assert sampler.sampler_context_cache is cache.global_context_cache and sampler.energy_context_cache is cache.global_context_cache
    sampler.sampler_context_cache = cc
    assert sampler.sampler_context_cache is cc and sampler.energy_context_cache is cache.global_context_cache


def test_sampler_state_subsets():
    """states.py:2296-2325 (tests/test_states.py __getitem__ cases): an index, a slice and an index list give independent copies
    with the same box and no energies."""
    rng = np.random.default_rng(0)
    box = np.diag([2.0, 3.0, 4.0])
    ss = states.SamplerState(rng.normal(size=(6, 3)), velocities=rng.normal(size=(6, 3)), box_vectors=box)
    ss.potential_energy = 1.0
    one, sl, pick = ss[2], ss[1:4], ss[[0, 5]]
    assert one.n_particles == 1 and np.array_equal(one.positions[0], ss.positions[2]) and np.array_equal(one.velocities[0], ss.velocities[2])
    assert sl.n_particles == 3 and np.array_equal(sl.positions, ss.positions[1:4])
    assert pick.n_particles == 2 and np.array_equal(pick.velocities, ss.velocities[[0, 5]])
    assert np.array_equal(sl.box_vectors, box) and sl.potential_energy is None and abs(ss.area_xy - 6.0) < 1e-15
    sl.positions[0, 0] += 1.0
    assert ss.positions[1, 0] != sl.positions[0, 0]
    assert states.SamplerState(np.zeros((2, 3)))[0].velocities is None and states.SamplerState(np.zeros((2, 3))).area_xy is None


def test_reduced_potential_at_states_is_the_energy_row_of_one_configuration():
    """states.py:144-183 (tests/test_states.py reduced_potential_at_states): temperatures on one System, and states on two
    Systems (two compatibility groups), against beta * U of the f64 oracle."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    from oracle.forcefield import ForceFieldOracle
    from openmmtools_amd.system import system_to_desc
    from openmmtools_amd.constants import kB
    a = testsystems.LennardJonesFluid(nparticles=216)
    b = testsystems.LennardJonesFluid(nparticles=216, epsilon=0.2 * unit.kilocalories_per_mole)
    ss = states.SamplerState(a.positions, box_vectors=a.system.getDefaultPeriodicBoxVectors())
    thermo = [states.ThermodynamicState(a.system, 100.0), states.ThermodynamicState(a.system, 150.0), states.ThermodynamicState(b.system, 100.0)]
    u = states.reduced_potential_at_states(ss, thermo, engine=OracleEngine(system_factory=ForceFieldOracle))
    box = np.diag(a.system.getDefaultPeriodicBoxVectors())
    Ua = ForceFieldOracle(system_to_desc(a.system)).energy_forces(a.positions, box)[0]
    Ub = ForceFieldOracle(system_to_desc(b.system)).energy_forces(a.positions, box)[0]
    assert np.allclose(u, [Ua / (kB * 100.0), Ua / (kB * 150.0), Ub / (kB * 100.0)], rtol=1e-10)


def test_alchemical_state_surface():
    """alchemy.py:86-262: the five parameters; set_alchemical_parameters moves the ones this state DEFINES (the bonded ones only where they
    were given, alchemy.py:94-99, 247-262); bonded lambdas on a System that names no softened terms are refused when the engine is set up."""
    a = states.AlchemicalState(lambda_sterics=0.5, lambda_electrostatics=0.25, lambda_bonds=1.0)
    assert (a.lambda_sterics, a.lambda_electrostatics, a.lambda_bonds, a.lambda_torsions) == (0.5, 0.25, 1.0, 1.0)
    a.set_alchemical_parameters(0.0)
    assert a.lambda_sterics == 0.0 and a.lambda_electrostatics == 0.0 and a.lambda_bonds == 1.0 and a.lambda_torsions == 1.0
    b = states.AlchemicalState(lambda_torsions=0.5)
    b.set_alchemical_parameters(0.25)
    assert (b.lambda_sterics, b.lambda_torsions, b.lambda_angles) == (0.25, 0.25, 1.0)
    with pytest.raises(ValueError):
        a.set_alchemical_parameters(1.5)
    from openmmtools_amd import alchemy, mcmc
    from openmmtools_amd.multistate import MultiStateSampler
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    from oracle.forcefield import ForceFieldOracle
    lj = testsystems.LennardJonesFluid(nparticles=64)
    asys = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, alchemy.AlchemicalRegion(alchemical_atoms=range(4)))
    th = [states.CompoundThermodynamicState(states.ThermodynamicState(asys, 120.0), [states.AlchemicalState(lambda_torsions=0.5)])]
    s = MultiStateSampler(mcmc_moves=mcmc.LangevinDynamicsMove(n_steps=1), number_of_iterations=1, engine=OracleEngine(ForceFieldOracle))
    with pytest.raises(NotImplementedError, match='lambda_bonds / lambda_angles / lambda_torsions'):
        s.create(th, [states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())], storage=None)
