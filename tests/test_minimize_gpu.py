"""GPU parity of the FIRE minimiser (integrate.hip: fire_move / fire_finish / fire_update kernels behind remd_minimize)
against the f64 oracle restatement of the reference's FIREMinimizationIntegrator (integrators.py:2290-2469).

FIRE is deterministic: the discrete protocol (accepted / restarted steps, time step, alpha) must follow the oracle step
for step as long as no energy difference is within fp32 noise of zero, and energies / positions agree to fp32 accuracy
over the short runs compared here."""
import numpy as np
import pytest
from openmmtools_amd import testsystems as ts, states, mcmc, unit
from openmmtools_amd.system import system_to_desc
from openmmtools_amd.multistate import ParallelTemperingSampler
from oracle import md_oracle as mo
from oracle.forcefield import ForceFieldOracle

pytestmark = pytest.mark.gpu
KB = 0.008314462618153242


def _setup(eng, system, x, R):
    desc = system_to_desc(system)
    eng.set_system(desc)
    eng.set_states(np.full(R, 1.0 / (KB * 300.0)))
    eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
    eng.seed(1)
    box = np.tile(np.diag(system.getDefaultPeriodicBoxVectors()), (R, 1))
    eng.set_replicas(R, 0, x, None, box, np.arange(R))
    return desc, box


def test_fire_lj_fluid_tracks_the_oracle(hip_engine_factory):
    lj = ts.LennardJonesFluid(nparticles=216)
    rng = np.random.default_rng(0)
    R = 3
    x = np.stack([lj.positions + 0.01 * r * rng.normal(size=lj.positions.shape) for r in range(R)])
    eng = hip_engine_factory()
    desc, box = _setup(eng, lj.system, x, R)
    u0 = eng.compute_energies(want_potential=True)[1]
    conv, n = eng.minimize(tolerance=0.0, max_iterations=30)
    assert n == 30 and not conv.any()
    xg, vg, ug, _ = eng.get_replicas(potential=True)
    ora = ForceFieldOracle(desc)
    for r in range(R):
        xo, vo, Eo, _, _ = mo.OracleFIRE(ora, tolerance=0.0).minimize(x[r], box[r], max_iterations=30)
        assert Eo < u0[r]
        assert abs(ug[r] - Eo) < 2e-4 * max(1.0, abs(Eo)) + 1e-3, (r, ug[r], Eo)
        d = xg[r] - xo
        d -= box[r] * np.round(d / box[r])
        assert np.abs(d).max() < 5e-4, np.abs(d).max()


def test_fire_converges_and_reports_it(hip_engine_factory):
    ho = ts.HarmonicOscillator()
    eng = hip_engine_factory()
    R = 4
    x = np.random.default_rng(2).normal(scale=0.05, size=(R, 1, 3))
    desc = system_to_desc(ho.system)
    eng.set_system(desc)
    eng.set_states(np.full(R, 1.0 / (KB * 300.0)))
    eng.set_integrator('V R O R V', 0.001, 1.0, 1, True, 1e-8)
    eng.set_replicas(R, 0, x, None, np.zeros((R, 3)), np.arange(R))
    conv, n = eng.minimize(tolerance=1e-2, max_iterations=0)
    assert conv.all() and 0 < n <= 2000
    xg, vg, ug, _ = eng.get_replicas(potential=True)
    assert np.abs(xg).max() < 1e-3 and np.all(ug < 1e-2)
    sysm = mo.OracleSystem(desc)
    _, _, _, c, n_o = mo.OracleFIRE(sysm, tolerance=1e-2).minimize(x[0])
    assert c and abs(n_o - n) <= 50                                   # the device polls convergence every 50 steps


def test_fire_alanine_with_constraints(hip_engine_factory):
    """Rigid water (SETTLE) + X-H SHAKE + PME: the minimiser lowers the energy, keeps every constraint, and follows the
    oracle's energy after a fixed number of steps."""
    al = ts.AlanineDipeptideExplicit()
    eng = hip_engine_factory()
    R = 2
    x = np.stack([al.positions, al.positions])
    desc, box = _setup(eng, al.system, x, R)
    u0 = eng.compute_energies(want_potential=True)[1]
    n_steps = 12
    conv, n = eng.minimize(tolerance=0.0, max_iterations=n_steps)
    xg, vg, ug, _ = eng.get_replicas(potential=True)
    assert np.all(ug < u0 - 50.0)                                     # kJ/mol: clearly downhill
    ora = ForceFieldOracle(desc)
    for (i, j, d0) in ora.constraints:
        d = xg[0][i] - xg[0][j]
        assert abs(np.linalg.norm(d) - d0) < 2e-6
    xo, vo, Eo, _, _ = mo.OracleFIRE(ora, tolerance=0.0).minimize(x[0], box[0], max_iterations=n_steps)
    assert abs(ug[0] - Eo) < 2e-5 * abs(Eo) + 0.5, (ug[0], Eo)


def test_sampler_minimize_on_device(hip_engine_factory):
    lj = ts.LennardJonesFluid(nparticles=216)
    tstate = states.ThermodynamicState(lj.system, 120.0)
    ss = states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, n_steps=10, reassign_velocities=True)
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=1, engine=hip_engine_factory(), seed=4)
    s.create(tstate, [ss], min_temperature=100.0, max_temperature=150.0, n_temperatures=3)
    s.run(0)
    s._compute_energies()
    before = s.energy_thermodynamic_states.diagonal().copy()
    conv, n = s.minimize(max_iterations=40)
    assert n == 40
    s.run()
    assert np.isfinite(s.energy_thermodynamic_states).all()
    x = np.stack([st.positions for st in s.sampler_states])
    assert np.isfinite(x).all() and not np.allclose(x[0], lj.positions)
