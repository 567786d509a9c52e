"""GPU: statistics of the HEADLINE system on the device: the 24-replica parallel-tempering ensemble of
AlanineDipeptideExplicit that bench.py times, 200 iterations of 500 g-BAOAB steps at 2 fs.

The long-run checks of tests/test_sampler_statistics_gpu.py are on the harmonic oscillator; here the system is the one with the PME
mesh, SETTLE / X-H constraints, the pair lists and the swap-all mix.  Known answers that need no second engine (the f64 oracle would
take hours on this size; the spirit is the 6-sigma tests of /root/reference/openmmtools/tests/test_sampling.py:287-307):

  kinetic temperature   after a propagation the replica in state k carries <KE> = n_dof k T_k / 2 with n_dof = 3 N - constraints - 3;
                        200 samples per state: standard error 0.15 %, bar 1 %
  constraints           every constrained distance of every replica at the end of the run, to the solver's relative tolerance
  neighbour swaps       the accepted / proposed counts between adjacent temperatures, summed over the run, against
                        E[min(1, exp((beta_k - beta_k+1)(U_a - U_b)))] over independent pairs U_a ~ state k, U_b ~ state k+1 -- the
                        acceptance a Metropolis swap between two canonical ensembles has -- estimated from the run's own potentials
  mixing                labels stay a permutation, every replica moves on the ladder
"""
import numpy as np
import pytest
from openmmtools_amd import testsystems, states, mcmc, unit
from openmmtools_amd.multistate import ParallelTemperingSampler

pytestmark = pytest.mark.gpu
KB = 0.008314462618153242


def test_headline_ensemble_temperatures_constraints_and_swap_rates(hip_engine_factory):
    al = testsystems.AlanineDipeptideExplicit()
    R, n_iter, n_eq = 24, 200, 10
    thermo = states.ThermodynamicState(al.system, 300.0 * unit.kelvin)
    ss = states.SamplerState(al.positions, box_vectors=al.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond, n_steps=500,
                                              reassign_velocities=True, splitting='V R R O R R V')
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=10 ** 9, engine=hip_engine_factory(), seed=0xC0FFEE)
    s.create(thermo, [ss], storage=None, min_temperature=300.0 * unit.kelvin, max_temperature=600.0 * unit.kelvin, n_temperatures=R)
    T = np.array([st.temperature for st in s._thermodynamic_states], dtype=float)
    beta = 1.0 / (KB * T)
    s.run(n_eq)                                                   # identical starting configurations: let the ladder spread first
    N = al.system.getNumParticles()
    n_con = al.system.getNumConstraints()
    n_dof = 3 * N - n_con - 3
    ke_sum, ke_n = np.zeros(R), np.zeros(R)
    U = [[] for _ in range(R)]                                    # potential energies sampled in each state
    acc = np.zeros((R, R)); prop = np.zeros((R, R))
    visited = np.zeros((R, R), dtype=bool)
    for _ in range(n_iter):
        s.run(1)
        lab = np.asarray(s._replica_thermodynamic_states)
        assert sorted(lab.tolist()) == list(range(R))             # swap-all keeps a permutation
        ke = np.asarray(s._engine.get_replicas(positions=False, velocities=False, kinetic=True)[3])
        u = s.energy_thermodynamic_states                          # [R, K] reduced potentials at the new positions
        for r in range(R):
            k = int(lab[r])
            ke_sum[k] += ke[r]; ke_n[k] += 1
            U[k].append(u[r, k] / beta[k])
            visited[r, k] = True
        acc += s._n_accepted_matrix; prop += s._n_proposed_matrix
    # kinetic temperatures
    T_kin = 2.0 * (ke_sum / ke_n) / (n_dof * KB)
    print('kinetic temperature / T_k: min %.4f max %.4f' % ((T_kin / T).min(), (T_kin / T).max()))
    assert np.abs(T_kin / T - 1.0).max() < 0.01, (T_kin / T)
    # every replica has been in several states (mixing along the ladder, not just permutation noise)
    assert visited.sum(axis=1).min() >= 4
    # constraints at the end of the run
    x = s._engine.get_replicas(positions=True, velocities=False)[0]
    box = np.asarray(s._engine.get_boxes()).reshape(R, 3)
    worst = 0.0
    for c in range(n_con):
        i, j, d0 = al.system.getConstraintParameters(c)
        d = x[:, j] - x[:, i]
        d -= box * np.round(d / box)
        worst = max(worst, float(np.abs(np.linalg.norm(d, axis=1) / d0 - 1.0).max()))
    assert worst < 2e-5, worst                                    # f32 positions: 1e-7 nm on 0.1 nm, Newton iterations of the X-H solver to 2e-7
    # neighbour swap acceptance against the two-ensemble expectation
    nsig_max = 0.0
    for k in range(R - 1):
        ua, ub = np.array(U[k]), np.array(U[k + 1])
        d = (beta[k] - beta[k + 1]) * (ua[:, None] - ub[None, :])
        p_pred = np.minimum(1.0, np.exp(np.minimum(d, 0.0))).mean()
        n_prop = prop[k, k + 1]
        assert n_prop > 0
        p_obs = acc[k, k + 1] / n_prop
        # the ~2 R proposals of one iteration share that iteration's energies: the independent sample is the iteration; the
        # prediction carries the error of two 200-sample ensembles as well
        sigma = np.sqrt(max(p_pred * (1.0 - p_pred), 0.01) / n_iter) * 2.0
        nsig_max = max(nsig_max, abs(p_obs - p_pred) / sigma)
    print('neighbour swap acceptance: largest deviation %.2f sigma; constraints: worst relative error %.2e' % (nsig_max, worst))
    assert nsig_max < 6.0, nsig_max
