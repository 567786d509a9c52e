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


@pytest.mark.parametrize('kind,world', [('alanine-pt', 2), ('lj-lambda', 2), ('lj-sams', 2), ('alanine-pt', 3)])
def test_hip_engine_sharded_equals_single_process(tmp_path, kind, world):
    """The HIP engine sharded over `world` ranks (sharing the one GPU of the test box, gloo rendezvous) reproduces the
    single-process HIP run BIT FOR BIT: labels, the gathered u_kl matrix, both count matrices every iteration, and each
    rank's positions and velocities at the end (reference: replicas distributed over ranks at
    multistatesampler.py:1296-1297, 1448-1449; rank-aware asserts tests/test_sampling.py:1233-1240).  Covers
    remd_set_replicas(r_begin > 0), device RNG keyed by the global replica index, and remd_mix on the gathered matrix."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, 'dist_worker_gpu.py')
    env = dict(os.environ, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run([sys.executable, worker, kind, str(tmp_path), 'single'], env=env, capture_output=True, text=True,
                         timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    port = 29700 + (os.getpid() % 200)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(port), worker, kind, str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    ref = np.load(os.path.join(tmp_path, 'ranksingle.npz'))
    ranks = [np.load(os.path.join(tmp_path, 'rank%d.npz' % r)) for r in range(world)]
    for z in ranks:                                    # every rank holds the full gathered state
        assert np.array_equal(z['labels'], ref['labels'])
        assert np.array_equal(z['ukl'], ref['ukl'])
        assert np.array_equal(z['nacc'], ref['nacc']) and np.array_equal(z['nprop'], ref['nprop'])
    assert np.isfinite(ref['ukl']).all()
    begins = [int(z['r_begin']) for z in ranks]
    assert begins[0] == 0 and all(b > 0 for b in begins[1:])
    assert np.array_equal(np.concatenate([z['x'] for z in ranks]), ref['x'])
    assert np.array_equal(np.concatenate([z['v'] for z in ranks]), ref['v'])
    assert len(np.unique(ref['labels'], axis=0)) > 1 or kind == 'lj-sams'    # something actually mixed
