"""Thermodynamic states on DIFFERENT Systems (several compatibility groups, states.py:186-217 / :994-1050): the reference's
own sampler test builds exactly that -- harmonic oscillators of different spring constants, one System per state
(tests/test_sampling.py:93-160) -- and keeps one Context per group (multistatesampler.py:1470-1490).  Here: one engine handle
per group behind the single-engine interface (openmmtools_amd/multistate/_engine_pool.py)."""
import numpy as np
import pytest
from openmmtools_amd import testsystems, states, mcmc, unit
from openmmtools_amd.constants import kB
from openmmtools_amd.multistate import MultiStateSampler, ReplicaExchangeSampler, SAMSSampler, MultiStateReporter
from openmmtools_amd.multistate import analysis as an
from openmmtools_amd.multistate._engine_pool import EnginePool
from oracle_engine import OracleEngine

T = 300.0
KT = kB * T                                                   # kJ/mol


def _oscillators(n_states=5):
    """tests/test_sampling.py:113-160: sigma_i = (1 + 0.2 i) A, K_i = kT / sigma_i^2, carbon mass; first and last are the
    unsampled end states; f_i = -3/2 ln(2 pi sigma_i^2)."""
    thermo, f_i, K_i = [], [], []
    for i in range(n_states + 2):
        sigma = 0.1 * (1.0 + 0.2 * i)                          # nm
        K = KT / sigma ** 2
        ho = testsystems.HarmonicOscillator(K=K * unit.kilojoules_per_mole / unit.nanometer ** 2, mass=12.0 * unit.amu)
        thermo.append(states.ThermodynamicState(ho.system, T * unit.kelvin))
        f_i.append(-1.5 * np.log(2.0 * np.pi * (sigma / 0.1) ** 2))
        K_i.append(K)
    ss = states.SamplerState(np.zeros((1, 3)), box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    return thermo[1:-1], [thermo[0], thermo[-1]], ss, np.array(f_i), np.array(K_i)


def _move(n_steps=25):
    return mcmc.LangevinSplittingDynamicsMove(timestep=4.0 * unit.femtosecond, collision_rate=20.0 / unit.picosecond,
                                              n_steps=n_steps, reassign_velocities=True, splitting='V R O R V')


def test_states_of_different_systems_are_grouped_and_get_a_handle_each():
    sampled, unsampled, ss, f_i, K_i = _oscillators()
    groups, idx = states.group_by_compatibility(sampled + unsampled)
    assert len(groups) == 7 and idx == [[k] for k in range(7)]                      # every spring constant is its own System
    same = states.ThermodynamicState(sampled[0].system, 350.0 * unit.kelvin)       # a temperature does not split a group
    groups, idx = states.group_by_compatibility(sampled + [same])
    assert len(groups) == 5 and idx[0] == [0, 5]
    s = ReplicaExchangeSampler(mcmc_moves=_move(), number_of_iterations=1, engine=OracleEngine(), seed=1)
    s.create(sampled + [same], [ss], storage=None)
    assert isinstance(s._engine, EnginePool) and s._engine.G == 5 and s._engine._groups[0] == [0, 5]


def test_energy_matrix_and_propagation_follow_each_states_own_system():
    """u_kl[r, k] = beta K_k |x_r|^2 / 2 with the spring constant of state k's OWN System (sampled and unsampled columns); a
    replica is propagated by the System of its current state: without mixing (MultiStateSampler) its positions spread with
    that state's sigma."""
    sampled, unsampled, ss, f_i, K_i = _oscillators()
    # collision rate 5 / ps here (near critical damping for these wells: positions decorrelate within ~0.5 ps; the 20 / ps of the
    # other tests is overdamped, relaxation time gamma / omega^2 up to 4 ps, and would leave a handful of independent samples)
    move = mcmc.LangevinSplittingDynamicsMove(timestep=4.0 * unit.femtosecond, collision_rate=5.0 / unit.picosecond, n_steps=120,
                                              reassign_velocities=True, splitting='V R O R V')
    n_iter = 240
    s = MultiStateSampler(mcmc_moves=move, number_of_iterations=n_iter, engine=OracleEngine(), seed=7)
    import copy
    s.create(sampled, [copy.deepcopy(ss) for _ in range(5)], storage=None, unsampled_thermodynamic_states=unsampled)
    x2 = np.zeros(5)
    for it in range(n_iter):
        s.run(1)
        x = np.stack([st.positions for st in s.sampler_states])[:, 0, :]
        r2 = (x ** 2).sum(axis=1)
        assert np.allclose(s.energy_thermodynamic_states, 0.5 * r2[:, None] * K_i[None, 1:-1] / KT, rtol=1e-12, atol=1e-12)
        assert np.allclose(s._energy_unsampled_states, 0.5 * r2[:, None] * K_i[None, [0, -1]] / KT, rtol=1e-12, atol=1e-12)
        x2 += r2 / 3.0
    sigma2 = KT / K_i[1:-1]
    assert list(s.replica_thermodynamic_states) == [0, 1, 2, 3, 4]
    # 115 ps per replica; measured autocorrelation time of |x|^2 ~ 1 ps (velocities are redrawn every 0.48 ps): ~100 independent
    # samples of a chi^2_3 variable, ~8 % standard error of the variance -- the bound is four of them
    assert np.all(np.abs(x2 / n_iter / sigma2 - 1.0) < 0.35), x2 / n_iter / sigma2


@pytest.mark.parametrize('cls', [ReplicaExchangeSampler, SAMSSampler])
def test_free_energies_of_the_oscillators_match_the_analytical_values(tmp_path, cls):
    """tests/test_sampling.py:93-300 (TestHarmonicOscillatorsMultiStateSampler and its replica-exchange / SAMS subclasses):
    f_j - f_i of the sampled and unsampled oscillators from MBAR on the stored energies against -3/2 ln(sigma_j^2 / sigma_i^2)."""
    sampled, unsampled, ss, f_i, K_i = _oscillators()
    kwargs = dict(mcmc_moves=_move(), number_of_iterations=300, engine=OracleEngine(), seed=11)
    s = cls(**kwargs)
    rep = MultiStateReporter(str(tmp_path / 'store'), checkpoint_interval=100)
    if cls is SAMSSampler:
        s.create(sampled, [ss], storage=rep, unsampled_thermodynamic_states=unsampled)
    else:
        s.create(sampled, [ss], storage=rep, unsampled_thermodynamic_states=unsampled)
    s.run()
    if cls is ReplicaExchangeSampler:
        assert s._n_accepted_matrix.sum() > 0 and sorted(s.replica_thermodynamic_states) == [0, 1, 2, 3, 4]
    D, dD = an.MultiStateSamplerAnalyzer(rep).get_free_energy()
    exact = f_i[None, :] - f_i[:, None]
    assert D.shape == (7, 7)
    err = np.abs(D - exact)
    assert np.all(err < 6.0 * dD + 1e-9), (err.max(), dD.max())
    assert 0.0 < dD[0, -1] < 0.4


def test_resume_keeps_the_groups(tmp_path):
    """from_storage rebuilds one handle per System from the stored states and continues the same trajectory."""
    sampled, unsampled, ss, f_i, K_i = _oscillators(3)
    def make(n):
        return ReplicaExchangeSampler(mcmc_moves=_move(10), number_of_iterations=n, engine=OracleEngine(), seed=3)
    a = make(6)
    a.create(sampled, [ss], storage=MultiStateReporter(str(tmp_path / 'a'), checkpoint_interval=1), unsampled_thermodynamic_states=unsampled)
    a.run()
    b = make(3)
    b.create(sampled, [ss], storage=MultiStateReporter(str(tmp_path / 'b'), checkpoint_interval=1), unsampled_thermodynamic_states=unsampled)
    b.run()
    del b
    r = ReplicaExchangeSampler.from_storage(str(tmp_path / 'b'), engine=OracleEngine())
    assert isinstance(r._engine, EnginePool) and r._engine.G == 5
    r.extend(3)
    assert r.iteration == 6 and list(r.replica_thermodynamic_states) == list(a.replica_thermodynamic_states)
    ea = MultiStateReporter(str(tmp_path / 'a'), open_mode='r').read_energies()[0]
    eb = MultiStateReporter(str(tmp_path / 'b'), open_mode='r').read_energies()[0]
    assert ea.shape == eb.shape == (7, 3, 3)
    assert np.allclose(ea[:4], eb[:4], rtol=0, atol=0)                     # the common part is the same run
    assert np.allclose(ea[4:], eb[4:], rtol=2e-5, atol=1e-6)               # the resumed part restarts from f4 checkpoints


def test_ghmc_statistics_are_kept_per_replica_across_the_group_handles():
    """GHMCMove over several compatibility groups: the Metropolis counters live per slot of each group's handle, the
    pool sums what each replica collected wherever it was propagated, and the sampler credits the moves of the states."""
    sampled, unsampled, ss, f_i, K_i = _oscillators(3)
    move = mcmc.GHMCMove(timestep=4.0 * unit.femtosecond, collision_rate=20.0 / unit.picosecond, n_steps=10)
    s = ReplicaExchangeSampler(mcmc_moves=move, number_of_iterations=6, engine=OracleEngine(), seed=3)
    s.create(sampled, [ss], storage=None)
    assert isinstance(s._engine, EnginePool) and s._engine.G == 3
    s.run()
    w = s._engine.get_work()
    assert w['n_trials'].tolist() == [60, 60, 60] and np.all(w['n_accepted'] <= w['n_trials']) and w['n_accepted'].sum() > 0
    moves = s.mcmc_moves
    assert sum(m.n_proposed for m in moves) == 180 and sum(m.n_accepted for m in moves) == int(w['n_accepted'].sum())


def test_particle_counts_must_agree():
    ho, two = testsystems.HarmonicOscillator(), testsystems.HarmonicOscillator()
    two.system.addParticle(12.0 * unit.amu)
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    s = MultiStateSampler(mcmc_moves=_move(), number_of_iterations=1, engine=OracleEngine(), seed=1)
    with pytest.raises(ValueError, match='same number of particles'):
        s.create([states.ThermodynamicState(ho.system, 300.0 * unit.kelvin), states.ThermodynamicState(two.system, 300.0 * unit.kelvin)],
                 [ss], storage=None)


@pytest.mark.gpu
def test_groups_on_the_device_follow_the_oracle_engines(hip_engine_factory):
    """The same replica-exchange run (five oscillators on five Systems + two unsampled) on device handles and on the f64
    oracle engines: identical labels every iteration, energies within the fp32 state of the device."""
    sampled, unsampled, ss, f_i, K_i = _oscillators()
    runs = []
    for eng in (hip_engine_factory(), OracleEngine()):
        s = ReplicaExchangeSampler(mcmc_moves=_move(20), number_of_iterations=12, engine=eng, seed=5)
        s.create(sampled, [ss], storage=None, unsampled_thermodynamic_states=unsampled)
        trace = []
        for it in range(12):
            s.run(1)
            trace.append((list(s.replica_thermodynamic_states), s.energy_thermodynamic_states.copy(), s._energy_unsampled_states.copy()))
        runs.append(trace)
        assert isinstance(s._engine, EnginePool) and s._engine.G == 7
    for (la, ua, uua), (lb, ub, uub) in zip(*runs):
        assert la == lb
        assert np.allclose(ua, ub, rtol=2e-4, atol=2e-4) and np.allclose(uua, uub, rtol=2e-4, atol=2e-4)


@pytest.mark.gpu
def test_lj_fluids_of_different_well_depths_exchange_on_the_device(hip_engine_factory):
    """A Hamiltonian ladder whose rungs are different Systems: LJ fluids (216 atoms) with epsilon scaled by 1, 0.9, 0.8 --
    three compatibility groups on three device handles.  Every column of u_kl is the f64 oracle's energy of the replica's
    positions in THAT column's System; replicas swap, and each one is propagated with the forces of its current System
    (its potential energy stays near the oracle's for the System it is in)."""
    from openmmtools_amd.system import system_to_desc
    from oracle.forcefield import ForceFieldOracle
    eps = [0.238 * s for s in (1.0, 0.9, 0.8)]
    fluids = [testsystems.LennardJonesFluid(nparticles=216, reduced_density=0.6, epsilon=e * unit.kilocalories_per_mole) for e in eps]
    thermo = [states.ThermodynamicState(f.system, 120.0 * unit.kelvin) for f in fluids]
    ss = states.SamplerState(fluids[0].positions, box_vectors=fluids[0].system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=5.0 / unit.picosecond, n_steps=20,
                                              reassign_velocities=True, splitting='V R O R V')
    s = ReplicaExchangeSampler(mcmc_moves=move, number_of_iterations=32, engine=hip_engine_factory(), seed=9)
    s.create(thermo, [ss], storage=None)
    s.minimize(max_iterations=60)
    assert isinstance(s._engine, EnginePool) and s._engine.G == 3
    oracles = [ForceFieldOracle(system_to_desc(f.system)) for f in fluids]
    box = np.diag(fluids[0].system.getDefaultPeriodicBoxVectors())
    beta = 1.0 / (kB * 120.0)
    seen = set()
    for it in range(32):                                                  # (a statistical statement: enough iterations for a swap whatever the noise)
        s.run(1)
        seen.add(tuple(s.replica_thermodynamic_states))
        if it >= 8 and len(seen) > 1:
            break
        x = np.stack([st.positions for st in s.sampler_states])
        ref = np.array([[beta * o.energy_forces(x[r], box)[0] for o in oracles] for r in range(3)])
        assert np.allclose(s.energy_thermodynamic_states, ref, rtol=1e-5, atol=1e-4), np.abs(s.energy_thermodynamic_states - ref).max()
    assert len(seen) > 1                                             # neighbouring well depths overlap: swaps are accepted


@pytest.mark.gpu
@pytest.mark.parametrize('system_name', ['LennardJonesFluid', 'AlanineDipeptideExplicit'])
def test_a_handle_holding_a_subset_keyed_by_global_indices_reproduces_those_replicas(hip_engine_factory, system_name):
    """remd_set_replica_ids (round 4): velocity reassignment and Langevin noise are keyed by what the host says a replica's global
    index is, so a handle that holds replicas {1, 3} of a four-replica ensemble (one handle per compatibility group,
    _engine_pool.py) moves them exactly as the handle that holds all four does -- on the resident small-system kernel (LJ fluid)
    and on the integrator chain (alanine dipeptide, constraints + PME)."""
    from openmmtools_amd.system import system_to_desc
    tsys = testsystems.LennardJonesFluid(nparticles=216) if system_name == 'LennardJonesFluid' else testsystems.AlanineDipeptideExplicit()
    desc = system_to_desc(tsys.system)
    box = np.diag(tsys.system.getDefaultPeriodicBoxVectors())
    rng = np.random.default_rng(5)
    x = np.stack([tsys.positions + 0.002 * rng.normal(size=tsys.positions.shape) for _ in range(4)])
    beta = 1.0 / (kB * np.array([300.0, 330.0, 360.0, 400.0]))
    out = {}
    for name, sel in (('all', [0, 1, 2, 3]), ('subset', [1, 3])):
        eng = hip_engine_factory()
        eng.set_system(desc)
        eng.set_states(beta)
        eng.set_integrator('V R O R V', 0.002 if system_name != 'LennardJonesFluid' else 0.001, 1.0, 20, True, 1e-8)
        eng.seed(11)
        eng.set_replicas(len(sel), 0, x[sel], None, np.tile(box, (len(sel), 1)), np.array(sel, dtype=np.int64))   # labels = own temperature
        if name == 'subset':
            eng.set_replica_ids(sel)
        eng.propagate(3)
        out[name] = eng.get_replicas()[:2]
    for k in range(2):
        assert np.array_equal(out['subset'][k], out['all'][k][[1, 3]])
    # and without the ids the subset is keyed 0, 1: another trajectory
    eng = hip_engine_factory()
    eng.set_system(desc); eng.set_states(beta)
    eng.set_integrator('V R O R V', 0.002 if system_name != 'LennardJonesFluid' else 0.001, 1.0, 20, True, 1e-8)
    eng.seed(11)
    eng.set_replicas(2, 0, x[[1, 3]], None, np.tile(box, (2, 1)), np.array([1, 3], dtype=np.int64))
    eng.propagate(3)
    assert not np.array_equal(eng.get_replicas()[0], out['subset'][0])


@pytest.mark.gpu
def test_copy_replicas_between_device_handles(hip_engine_factory):
    """remd_copy_replicas: a handle sized with x = NULL and filled device-to-device evaluates and propagates bit for bit like a
    handle that was given the same coordinates from the host (solvated dipeptide: molecule sort, cluster lists, mesh)."""
    from openmmtools_amd import testsystems
    from openmmtools_amd.system import system_to_desc
    al = testsystems.AlanineDipeptideExplicit()
    desc = system_to_desc(al.system, ewald_split='auto')
    box = np.tile(np.diag(al.system.getDefaultPeriodicBoxVectors()), (4, 1))
    rng = np.random.default_rng(2)
    x = np.stack([al.positions + 0.002 * rng.normal(size=al.positions.shape) for _ in range(4)])
    v = 0.3 * rng.normal(size=x.shape)
    a, b, c = hip_engine_factory(), hip_engine_factory(), hip_engine_factory()
    for eng in (a, b, c):
        eng.set_system(desc)
        eng.set_states(np.full(4, 1.0 / (0.008314462618153242 * 300.0)))
        eng.set_integrator('V R O R V', 0.002, 1.0, 10, False, 1e-8)
        eng.seed(9)
    a.set_replicas(4, 0, x, v, box, np.arange(4))
    a.compute_energies()                                   # (the source has work in flight on its own stream)
    b.set_replicas(2, 0, None, None, box[:2], np.arange(2))
    b.copy_replicas([1, 0], a, [3, 1], 7)
    xa, va, _, _ = a.get_replicas()
    xb, vb, _, _ = b.get_replicas()
    assert np.array_equal(xb, xa[[1, 3]]) and np.array_equal(vb, va[[1, 3]])
    c.set_replicas(2, 0, xa[[1, 3]], va[[1, 3]], box[:2], np.arange(2))
    for eng in (b, c):
        eng.set_replica_ids([1, 3])
    assert np.array_equal(b.compute_energies(), c.compute_energies()) and np.array_equal(b.get_forces(), c.get_forces())
    assert not b.propagate(0).any() and not c.propagate(0).any()
    for got, ref in zip(b.get_replicas()[:2], c.get_replicas()[:2]):
        assert np.array_equal(got, ref)
    a.copy_replicas([3, 1], b, [0, 1], 3)                  # and back into the master
    xa2 = a.get_replicas()[0]
    assert np.array_equal(xa2[[3, 1]], b.get_replicas()[0]) and np.array_equal(xa2[[0, 2]], xa[[0, 2]])


def test_group_by_compatibility_is_the_references_function():
    """states.py:186-217 executed from the reference's source on stand-in states (tests/golden/make_golden_group_by_compatibility.py):
    same groups, same order, same original indices."""
    import json
    import os
    from openmmtools_amd import states as st
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'group_by_compatibility_reference.json')))

    class State:
        def __init__(self, kind, index):
            self.kind, self.index = kind, index

        def is_state_compatible(self, other):
            return self.kind == other.kind
    for c in G['cases']:
        groups, indices = st.group_by_compatibility([State(k, i) for i, k in enumerate(c['kinds'])])
        assert [[s.index for s in g] for g in groups] == c['groups'] and [list(i) for i in indices] == c['original_indices'], c
