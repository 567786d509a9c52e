"""Worker for tests/test_distributed_gpu.py::test_hip_engine_sharded_*: the HIP engine under torch.distributed with
world_size ranks SHARING ONE GPU (gloo rendezvous; RCCL refuses two ranks on one device) — the sharded code path of the
product (remd_set_replicas with r_begin > 0, global-replica RNG keying on the device, remd_mix on the gathered device
matrix with a leading dimension) against the single-process HIP run.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/dist_worker_gpu.py <kind> <outdir>
    python tests/dist_worker_gpu.py <kind> <outdir> single
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

N_ITER = 3


def build(kind, engine, comm):
    from openmmtools_amd import testsystems, states, mcmc, unit, alchemy
    from openmmtools_amd.multistate import ParallelTemperingSampler, ReplicaExchangeSampler, SAMSSampler
    if kind == 'alanine-pt':                       # PME + SETTLE/SHAKE + CM-motion removal, 6 temperatures, swap-all
        al = testsystems.AlanineDipeptideExplicit()
        move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                                  n_steps=30, reassign_velocities=True, splitting='V R R O R R V')
        s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=N_ITER, engine=engine, seed=77, comm=comm,
                                     online_analysis_interval=None)
        ts = states.ThermodynamicState(al.system, 300.0)
        ss = states.SamplerState(al.positions, box_vectors=al.system.getDefaultPeriodicBoxVectors())
        s.create(ts, [ss], storage=None, min_temperature=300.0, max_temperature=450.0, n_temperatures=6)
    elif kind == 'lj-lambda':                      # 5 lambda_sterics states on 5 replicas (ragged blocks 3 + 2), swap-all
        lj = testsystems.LennardJonesFluid(nparticles=216)
        region = alchemy.AlchemicalRegion(alchemical_atoms=range(6))
        asys = alchemy.AbsoluteAlchemicalFactory().create_alchemical_system(lj.system, region)
        ths = [states.CompoundThermodynamicState(states.ThermodynamicState(asys, 120.0),
                                                 [states.AlchemicalState(lambda_sterics=l, lambda_electrostatics=1.0)])
               for l in np.linspace(1.0, 0.0, 5)]
        move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                                  n_steps=40, reassign_velocities=False, splitting='V R O R V')
        s = ReplicaExchangeSampler(mcmc_moves=move, number_of_iterations=N_ITER, engine=engine, seed=5, comm=comm,
                                   online_analysis_interval=None)
        s.create(ths, [states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())], storage=None)
    elif kind == 'lj-sams':                        # 4 replicas jumping among 7 temperature states
        lj = testsystems.LennardJonesFluid(nparticles=216)
        ths = [states.ThermodynamicState(lj.system, T) for T in np.linspace(100.0, 160.0, 7)]
        move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                                  n_steps=40, reassign_velocities=True, splitting='V R O R V')
        s = SAMSSampler(mcmc_moves=move, number_of_iterations=N_ITER, engine=engine, seed=9, comm=comm,
                        flatness_criteria='minimum-visits', online_analysis_interval=None)
        s.create(ths, [states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())] * 4,
                 storage=None)
    else:
        raise ValueError(kind)
    s.verify_labels = comm is not None
    return s


def run(kind, comm):
    from openmmtools_amd._engine import HipEngine
    s = build(kind, HipEngine(device=0), comm)
    hist = []
    for _ in range(N_ITER):
        s.run(1)
        hist.append((s.replica_thermodynamic_states.copy(), s.energy_thermodynamic_states.copy(),
                     s._n_accepted_matrix.copy(), s._n_proposed_matrix.copy()))
    x, v, _, _ = s._engine.get_replicas()          # this rank's block, as it sits on the device
    return hist, x, v, s._r_begin, s._r_count


if __name__ == '__main__':
    kind, out = sys.argv[1], sys.argv[2]
    single = len(sys.argv) > 3 and sys.argv[3] == 'single'
    import torch
    torch.cuda.set_device(0)
    if single:
        comm, rank = None, 'single'
    else:
        import torch.distributed as dist
        from openmmtools_amd.multistate.comm import TorchDistributedComm
        dist.init_process_group('gloo')
        comm = TorchDistributedComm()
        rank = comm.rank
    hist, x, v, b, c = run(kind, comm)
    np.savez(os.path.join(out, 'rank%s.npz' % rank), labels=np.stack([h[0] for h in hist]),
             ukl=np.stack([h[1] for h in hist]), nacc=np.stack([h[2] for h in hist]), nprop=np.stack([h[3] for h in hist]),
             x=x, v=v, r_begin=b, r_count=c)
    if not single:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
