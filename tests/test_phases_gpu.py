"""GPU: a sampler whose engine propagates its replicas as two phases (include/remd_hip.h: remd_set_phases) is the same sampler.

The reference propagates the replicas one after the other or one per MPI rank (multistate/multistatesampler.py:1296-1297): they are
independent between two mixes, so HOW the device schedules them must not show in any result.  Here the 16-replica parallel-tempering
ensemble of AlanineDipeptideExplicit (the bench.py system) runs four iterations -- mix, propagate, energy matrix, with swap-all
accepting and rejecting on the way -- once as one block and once as two blocks whose MD steps take turns: the labels, the count
matrices, every reduced potential and every coordinate must be identical, bit for bit."""
import numpy as np
import pytest
from openmmtools_amd import testsystems, states, mcmc, unit
from openmmtools_amd.multistate import ParallelTemperingSampler

pytestmark = pytest.mark.gpu


def _run(engine, phases, R=16, n_iter=4):
    al = testsystems.AlanineDipeptideExplicit()
    engine.set_phases(phases)
    thermo = states.ThermodynamicState(al.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(al.positions, box_vectors=al.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond, n_steps=60,
                                              reassign_velocities=True, splitting='V R R O R R V')
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=10 ** 9, engine=engine, seed=0xC0FFEE)
    s.create(thermo, [ss], storage=None, min_temperature=300.0 * unit.kelvin, max_temperature=320.0 * unit.kelvin, n_temperatures=R)
    out = []
    for _ in range(n_iter):
        s.run(1)
        x, v = engine.get_replicas()[:2]
        out.append((np.array(s._replica_thermodynamic_states), np.array(s.energy_thermodynamic_states), np.array(s._n_accepted_matrix),
                    np.array(s._n_proposed_matrix), x, v))
    return out, engine.phases_active()


def test_a_sampler_on_two_phases_is_the_sampler_on_one_block(hip_engine_factory):
    one, p1 = _run(hip_engine_factory(), 1)
    two, p2 = _run(hip_engine_factory(), 2)
    assert (p1, p2) == (1, 2)
    assert any(not np.array_equal(a[0], np.arange(16)) for a in one), 'the ladder never exchanged: the test would not see a label bug'
    for it, (a, b) in enumerate(zip(one, two)):
        for q, name in enumerate(('labels', 'u_kl', 'accepted', 'proposed', 'positions', 'velocities')):
            assert np.array_equal(a[q], b[q]), (it, name)


def test_phases_by_rule_follow_the_hardware_queue_limit(hip_engine_factory, monkeypatch):
    """By rule (remd_set_phases(0), the default) a handle runs two blocks only when the process keeps its streams on few hardware queues
    (GPU_MAX_HW_QUEUES = 2 (or 3), set by the package before the HIP runtime starts) and holds 6 replicas or more for 16 MD steps or more; REMD_PHASES=1 pins one
    block.  (What the limit buys is in profiles/r06_phases_hw_queues.txt; here only the rule.)"""
    import os
    from openmmtools_amd.system import system_to_desc
    al = testsystems.AlanineDipeptideExplicit()
    desc = system_to_desc(al.system, ewald_split='auto')
    box = np.diag(al.system.getDefaultPeriodicBoxVectors())

    def phases_of(R, env, n_steps=16):
        for k in ('GPU_MAX_HW_QUEUES', 'REMD_PHASES'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = hip_engine_factory()
        eng.set_system(desc); eng.set_states(np.full(R, 1.0 / (0.008314462618153242 * 300.0)))
        eng.set_integrator('V R R O R R V', 0.002, 1.0, n_steps, True, 1e-8)
        eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
        eng.propagate(0)
        return eng.phases_active()
    assert phases_of(16, {'GPU_MAX_HW_QUEUES': '2'}) == 2
    assert phases_of(16, {}) == 1 and phases_of(16, {'GPU_MAX_HW_QUEUES': '4'}) == 1 and phases_of(16, {'GPU_MAX_HW_QUEUES': '1'}) == 1
    assert phases_of(8, {'GPU_MAX_HW_QUEUES': '2'}) == 2 and phases_of(4, {'GPU_MAX_HW_QUEUES': '2'}) == 1
    assert phases_of(16, {'GPU_MAX_HW_QUEUES': '2', 'REMD_PHASES': '1'}) == 1
    assert phases_of(16, {'GPU_MAX_HW_QUEUES': '2'}, n_steps=8) == 1              # (short propagations: the set-up of the blocks costs more than they save)


def test_a_monte_carlo_barostat_in_two_phases_is_the_barostat_in_one_block(hip_engine_factory):
    """NPT states: every block runs the volume moves of its own replicas (per-replica volume step, adaptation window and totals, the
    handle's step and attempt counters travel with the replicas; the draws are Philox by global replica and attempt).  Positions,
    velocities, boxes and the barostat's statistics after three propagations of 12 replicas (a move every 5 steps) equal the one-block
    run bit for bit."""
    from openmmtools_amd.system import system_to_desc
    al = testsystems.AlanineDipeptideExplicit()
    desc = system_to_desc(al.system, ewald_split='auto')
    box = np.diag(al.system.getDefaultPeriodicBoxVectors())
    R = 12
    KB = 0.008314462618153242
    out = []
    for phases in (1, 2):
        eng = hip_engine_factory()
        eng.set_phases(phases)
        eng.set_system(desc)
        T = np.linspace(300.0, 330.0, R)
        eng.set_states(1.0 / (KB * T))
        eng.set_barostat(np.full(R, 0.0602214076 * 1.01325), frequency=5)          # 1 atm in kJ/mol/nm^3
        eng.set_integrator('V R R O R R V', 0.002, 1.0, 30, True, 1e-8)
        eng.seed(21)
        eng.set_replicas(R, 0, np.tile(al.positions, (R, 1, 1)), None, np.tile(box, (R, 1)), np.arange(R))
        for it in range(3):
            assert not eng.propagate(it).any()
            u = eng.compute_energies()           # (as a sampler does between two propagations: the forces of the first kick come from here)
        x, v = eng.get_replicas()[:2]
        out.append((x.copy(), v.copy(), eng.get_boxes(), eng.barostat_stats(), u, eng.phases_active()))
    (xa, va, ba, sa, ua, pa), (xb, vb, bb, sb, ub, pb) = out
    assert (pa, pb) == (1, 2)
    assert sa[1].min() == 18 and np.any(sa[2] > 0) and not np.allclose(ba, box)       # 3 x 30 / 5 attempts per replica, some accepted, boxes moved
    assert np.array_equal(ba, bb) and all(np.array_equal(p, q) for p, q in zip(sa, sb))
    assert np.array_equal(xa, xb) and np.array_equal(va, vb) and np.array_equal(ua, ub)


@pytest.mark.parametrize('which', ['HostGuestVacuum', 'AlanineDipeptideImplicit'])
def test_nocutoff_systems_in_two_phases_are_the_one_block_run(hip_engine_factory, which):
    """NoCutoff systems beyond the resident small-molecule kernel (more than 64 atoms, or with GBSA): a step is three to four dependent small
    launches on one stream; such a handle runs as two blocks on two streams when asked to (remd_set_phases(2); the implicit solvent's
    descriptor travels to the blocks: remd_gbsa_clone) -- bit-identical to one block, and no faster: the host's enqueue rate is the limit,
    so the rule keeps these handles in one block."""
    from openmmtools_amd.system import system_to_desc
    t = getattr(testsystems, which)()
    desc = system_to_desc(t.system)
    R = 8
    out = []
    for phases in (1, 2):
        eng = hip_engine_factory()
        eng.set_phases(phases)
        eng.set_system(desc)
        eng.set_states(1.0 / (0.008314462618153242 * np.linspace(300.0, 400.0, R)))
        eng.set_integrator('V R R O R R V', 0.002, 1.0, 60, True, 1e-8)
        eng.seed(5)
        eng.set_replicas(R, 0, np.tile(t.positions, (R, 1, 1)), None, np.zeros((R, 3)), np.arange(R))
        for it in range(2):
            assert not eng.propagate(it).any()
            u = eng.compute_energies()
        x, v = eng.get_replicas()[:2]
        out.append((x.copy(), v.copy(), u.copy(), eng.phases_active()))
    (xa, va, ua, pa), (xb, vb, ub, pb) = out
    assert (pa, pb) == (1, 2)
    assert np.array_equal(xa, xb) and np.array_equal(va, vb) and np.array_equal(ua, ub)


def test_new_boxes_reach_the_blocks(hip_engine_factory):
    """The blocks of a phased handle keep their own box mirrors and PME influence tables: replicas set again with OTHER boxes (same shapes,
    so the blocks are not re-made) must be propagated in the new boxes -- against the one-block run, bit for bit."""
    from openmmtools_amd.system import system_to_desc
    al = testsystems.AlanineDipeptideExplicit()
    desc = system_to_desc(al.system, ewald_split='auto')
    box = np.diag(al.system.getDefaultPeriodicBoxVectors())
    R = 4
    x0 = np.tile(al.positions, (R, 1, 1))
    out = []
    for phases in (1, 2):
        eng = hip_engine_factory()
        eng.set_phases(phases)
        eng.set_system(desc); eng.set_states(np.full(R, 1.0 / (0.008314462618153242 * 300.0)))
        eng.set_integrator('V R R O R R V', 0.002, 1.0, 20, True, 1e-8)
        eng.seed(7)
        eng.set_replicas(R, 0, x0, None, np.tile(box, (R, 1)), np.arange(R))
        eng.propagate(0)
        x, v = eng.get_replicas()[:2]
        eng.set_replicas(R, 0, x * 1.002, v, np.tile(box * 1.002, (R, 1)), np.arange(R))      # an isotropic rescale, as a barostat would make
        eng.propagate(1)
        out.append(eng.get_replicas()[:2] + (eng.compute_energies(),))
    for q in range(3):
        assert np.array_equal(out[0][q], out[1][q]), q
