"""GPU: the multi-rank code path (torch tensor u_kl rows -> RCCL all-gather -> device mix on the gathered matrix with
a leading dimension) exercised with a world-size-1 NCCL group on the single GPU of the test box, and checked against
the single-process path.  (The 2/4/8-GPU runs are the driver's; sharding logic itself is covered on CPU with gloo.)"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_nccl_world1_path_matches_single_process_path(hip_engine_factory):
    import torch
    import torch.distributed as dist
    from openmmtools_amd import testsystems, states, mcmc, unit
    from openmmtools_amd.multistate import ParallelTemperingSampler
    from openmmtools_amd.multistate.comm import TorchDistributedComm
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    torch.cuda.set_device(0)
    created = False
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        created = True
    try:
        lj = testsystems.LennardJonesFluid(nparticles=216)
        ts = states.ThermodynamicState(lj.system, 120.0)
        ss = states.SamplerState(lj.positions, box_vectors=lj.system.getDefaultPeriodicBoxVectors())
        move = mcmc.LangevinSplittingDynamicsMove(timestep=1.0 * unit.femtosecond, collision_rate=1.0 / unit.picosecond,
                                                  n_steps=20, reassign_velocities=True, splitting='V R O R V')
        results = []
        for comm in (None, TorchDistributedComm()):
            s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=4, engine=hip_engine_factory(), seed=5, comm=comm)
            s.create(ts, [ss], min_temperature=100.0, max_temperature=200.0, n_temperatures=6)
            s.verify_labels = comm is not None
            s.run()
            results.append((s.replica_thermodynamic_states.copy(), s.energy_thermodynamic_states.copy(),
                            s._n_proposed_matrix.copy(), s._n_accepted_matrix.copy()))
        for a, b in zip(results[0], results[1]):
            assert np.array_equal(a, b)
        assert np.isfinite(results[0][1]).all()
    finally:
        if created:
            dist.destroy_process_group()
