"""Worker for tests/test_distributed_cpu.py: runs the sharded sampler under torch.distributed (gloo)."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def build(sampler_kind, engine, comm, n_iter, storage=None):
    from openmmtools_amd import testsystems, states, mcmc, unit
    from openmmtools_amd.multistate import ParallelTemperingSampler, SAMSSampler
    ho = testsystems.HarmonicOscillator()
    ts = states.ThermodynamicState(ho.system, 300.0)
    ss = states.SamplerState(ho.positions, box_vectors=ho.system.getDefaultPeriodicBoxVectors())
    move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                              n_steps=25, reassign_velocities=True, splitting='V R O R V')
    if sampler_kind == 'regions':
        # general alchemical regions (tests/test_alchemical_regions.py): four states of two named regions on a charged fluid, the engine =
        # the C++ build of the C ABI; the regions' tables live per handle, the replicas' own states follow the GLOBAL labels
        from test_alchemical_regions import _two_region_sampler
        from openmmtools_amd.multistate import ReplicaExchangeSampler
        s, _ = _two_region_sampler(engine, storage, n_iter, comm=comm)
        s.verify_labels = True
        return s
    if sampler_kind == 'groups':
        # four oscillators of different spring constants = four Systems = four compatibility groups (tests/test_compat_groups.py)
        from openmmtools_amd.multistate import ReplicaExchangeSampler
        from openmmtools_amd.constants import kB
        sts = []
        for i in range(4):
            K = kB * 300.0 / (0.1 * (1.2 + 0.2 * i)) ** 2
            sts.append(states.ThermodynamicState(testsystems.HarmonicOscillator(
                K=K * unit.kilojoules_per_mole / unit.nanometer ** 2, mass=12.0 * unit.amu).system, 300.0))
        s = ReplicaExchangeSampler(mcmc_moves=move, number_of_iterations=n_iter, engine=engine, seed=77, comm=comm)
        s.create(sts, [ss], storage=storage)
    elif sampler_kind == 'mc':
        # a Metropolized displacement, a GHMC integration and a Langevin integration per iteration (mcmc.py:810-975, 1323-1490):
        # host proposals keyed by the GLOBAL replica, the engine reprogrammed between the two integrator moves
        from openmmtools_amd.multistate import ReplicaExchangeSampler
        seq = mcmc.SequenceMove([mcmc.MCDisplacementMove(displacement_sigma=0.05 * unit.nanometer),
                                 mcmc.GHMCMove(timestep=2.0 * unit.femtosecond, n_steps=6), move])
        sts = [states.ThermodynamicState(ho.system, T) for T in np.linspace(300.0, 450.0, 5)]
        s = ReplicaExchangeSampler(mcmc_moves=seq, number_of_iterations=n_iter, engine=engine, seed=77, comm=comm,
                                   online_analysis_interval=None)
        s.create(sts, [ss], storage=storage)
    elif sampler_kind == 'pt':
        s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=n_iter, engine=engine, seed=77, comm=comm,
                                     online_analysis_interval=3)       # MBAR on rank 0 at iterations 3 and 6, error broadcast
        s.create(ts, [ss], storage=storage, min_temperature=300.0, max_temperature=600.0, n_temperatures=5)
    else:
        sts = [states.ThermodynamicState(ho.system, T) for T in np.linspace(300.0, 500.0, 6)]
        s = SAMSSampler(mcmc_moves=move, number_of_iterations=n_iter, engine=engine, seed=77, comm=comm,
                        flatness_criteria='minimum-visits')
        s.create(sts, [ss] * 4, storage=storage)
    s.verify_labels = True
    return s


def run(sampler_kind, comm, n_iter=6, storage_dir=None):
    from oracle_engine import OracleEngine
    from openmmtools_amd.multistate import MultiStateReporter
    storage = MultiStateReporter(os.path.join(storage_dir, 'store'), checkpoint_interval=2) if storage_dir else None
    if sampler_kind == 'regions':
        import oracle
        from openmmtools_amd._engine import HipEngine
        class CpuBuild(HipEngine):            # the C++ build of the ABI: rows come back on the host (gloo all-gather of host rows)
            is_device = False
        engine = CpuBuild(lib_path=os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so'))
    else:
        engine = OracleEngine()
    s = build(sampler_kind, engine, comm, n_iter, storage)
    history = []
    analysis = []
    for _ in range(n_iter):
        s.run(1)
        history.append((s.replica_thermodynamic_states.copy(), s.energy_thermodynamic_states.copy(),
                        s._n_accepted_matrix.copy(), s._n_proposed_matrix.copy()))
        analysis.append(np.append(s._last_mbar_f_k, s._last_err_free_energy) if sampler_kind not in ('groups', 'mc', 'regions') else np.zeros(1))
    x = np.stack([st.positions for st in s.sampler_states])
    run.last_analysis = np.stack(analysis)          # [iteration, K + 1]: online f_k and the current error estimate
    return history, x, (s._r_begin, s._r_count)


if __name__ == '__main__':
    import torch.distributed as dist
    from openmmtools_amd.multistate.comm import TorchDistributedComm
    kind, out = sys.argv[1], sys.argv[2]
    dist.init_process_group('gloo')
    comm = TorchDistributedComm()
    if kind == 'exists':
        # the store under <out>/store was created by the test: create() must refuse on EVERY rank, before any collective
        from oracle_engine import OracleEngine
        from openmmtools_amd.multistate import MultiStateReporter
        try:
            build('pt', OracleEngine(), comm, 2, MultiStateReporter(os.path.join(out, 'store'), checkpoint_interval=2))
            verdict = 'created'
        except RuntimeError as e:
            verdict = 'refused' if 'refusing to overwrite' in str(e) else 'other: %s' % e
        with open(os.path.join(out, 'exists_rank%d.txt' % comm.rank), 'w') as fh:
            fh.write(verdict)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0)
    history, x, (b, c) = run(kind, comm, storage_dir=out)
    np.savez(os.path.join(out, 'rank%d.npz' % comm.rank),
             labels=np.stack([h[0] for h in history]), ukl=np.stack([h[1] for h in history]),
             nacc=np.stack([h[2] for h in history]), nprop=np.stack([h[3] for h in history]),
             x_local=x[b:b + c], r_begin=b, r_count=c, analysis=run.last_analysis)
    dist.barrier()
    dist.destroy_process_group()
