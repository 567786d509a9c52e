"""world_size-2 gloo tests of the N > 1 path on CPU: replica sharding, all-gather of u_kl rows, replicated
deterministic mixing (+ label consistency broadcast).  Because every random stream is keyed by the GLOBAL
replica index, the sharded run must reproduce the single-process run exactly."""
import os
import subprocess
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def test_block_partition():
    from openmmtools_amd.multistate.comm import block_partition
    assert block_partition(24, 8) == ([0, 3, 6, 9, 12, 15, 18, 21], [3] * 8)
    assert block_partition(5, 2) == ([0, 3], [3, 2])
    assert block_partition(3, 4) == ([0, 1, 2, 3], [1, 1, 1, 0])


@pytest.mark.parametrize('kind', ['pt', 'sams', 'mc', 'regions'])
def test_sharded_run_equals_single_process(tmp_path, kind):
    import dist_worker
    from openmmtools_amd.multistate.comm import SingleProcessComm
    os.makedirs(tmp_path / 'single')
    ref_hist, ref_x, _ = dist_worker.run(kind, SingleProcessComm(), storage_dir=str(tmp_path / 'single'))
    ref_analysis = dist_worker.run.last_analysis.copy()
    port = 29600 + (os.getpid() % 200) + ['pt', 'sams', 'mc', 'regions'].index(kind)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(HERE, 'dist_worker.py'), kind,
           str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    ranks = [np.load(os.path.join(tmp_path, 'rank%d.npz' % r)) for r in range(2)]
    for it, (labels, ukl, nacc, nprop) in enumerate(ref_hist):
        for z in ranks:                                    # every rank holds the full gathered state
            assert np.array_equal(z['labels'][it], labels)
            assert np.array_equal(z['ukl'][it], ukl)       # bit-identical f64 rows after the all-gather
            assert np.array_equal(z['nacc'][it], nacc) and np.array_equal(z['nprop'][it], nprop)
    # online analysis: replicated stochastic-approximation estimate; the MBAR error (rank 0, iterations 3 and 6) is broadcast
    for z in ranks:
        assert np.array_equal(z['analysis'], ref_analysis, equal_nan=True)
    # positions never leave their rank: each rank's block equals the matching block of the reference run
    got = np.concatenate([z['x_local'] for z in ranks])
    assert np.array_equal(got, ref_x)
    assert int(ranks[0]['r_begin']) == 0 and int(ranks[1]['r_begin']) == int(ranks[0]['r_count'])
    # storage: rank 0 alone writes; the checkpoint gathers both ranks' blocks and equals the single-process store
    from openmmtools_amd.multistate import MultiStateReporter
    a = MultiStateReporter(os.path.join(tmp_path, 'store'), open_mode='r')
    b = MultiStateReporter(os.path.join(tmp_path, 'single', 'store'), open_mode='r')
    assert a.read_checkpoint_iterations() == b.read_checkpoint_iterations() == [0, 2, 4, 6]
    for it in (2, 6):
        xa = np.stack([s.positions for s in a.read_sampler_states(it)])
        xb = np.stack([s.positions for s in b.read_sampler_states(it)])
        assert np.array_equal(xa, xb)
    assert np.array_equal(a.read_energies()[0], b.read_energies()[0])
    assert np.array_equal(a.read_replica_thermodynamic_states(), b.read_replica_thermodynamic_states())


def test_sharded_run_with_several_compatibility_groups(tmp_path):
    """States on different Systems (one engine handle per group on every rank, multistate/_engine_pool.py) under two ranks:
    every rank holds the same gathered energy matrix and labels, swaps happen, and the stored energies are those of the stored
    positions in every state's own System (u = beta K_k |x|^2 / 2).  Round 4: a group's batch is keyed by the replicas' GLOBAL
    indices (remd_set_replica_ids), so the two-rank run IS the single-process trajectory -- labels, energy matrix, counts and
    positions bit for bit -- like a single-group run."""
    from openmmtools_amd.constants import kB
    import dist_worker
    from openmmtools_amd.multistate.comm import SingleProcessComm
    os.makedirs(tmp_path / 'single')
    ref_hist, ref_x, _ = dist_worker.run('groups', SingleProcessComm(), storage_dir=str(tmp_path / 'single'))
    port = 29850 + (os.getpid() % 100)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(HERE, 'dist_worker.py'), 'groups', str(tmp_path)]
    res = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS='1'), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    ranks = [np.load(os.path.join(tmp_path, 'rank%d.npz' % r)) for r in range(2)]
    assert np.array_equal(ranks[0]['labels'], ranks[1]['labels']) and np.array_equal(ranks[0]['ukl'], ranks[1]['ukl'])
    assert np.array_equal(ranks[0]['nacc'], ranks[1]['nacc']) and ranks[0]['nacc'].sum() > 0
    assert all(sorted(l) == [0, 1, 2, 3] for l in ranks[0]['labels'])
    for it, (labels, ukl, nacc, nprop) in enumerate(ref_hist):           # rank-count invariance
        assert np.array_equal(ranks[0]['labels'][it], labels) and np.array_equal(ranks[0]['ukl'][it], ukl)
        assert np.array_equal(ranks[0]['nacc'][it], nacc) and np.array_equal(ranks[0]['nprop'][it], nprop)
    assert np.array_equal(np.concatenate([z['x_local'] for z in ranks]), ref_x)
    from openmmtools_amd.multistate import MultiStateReporter
    rep = MultiStateReporter(os.path.join(tmp_path, 'store'), open_mode='r')
    K = np.array([kB * 300.0 / (0.1 * (1.2 + 0.2 * i)) ** 2 for i in range(4)])
    e = rep.read_energies()[0]
    for it in (2, 4, 6):
        x = np.stack([s.positions for s in rep.read_sampler_states(it)])[:, 0, :]
        assert np.allclose(e[it], 0.5 * (x ** 2).sum(axis=1)[:, None] * K[None, :] / (kB * 300.0), rtol=2e-5, atol=1e-7)
    assert np.array_equal(e[6], ranks[0]['ukl'][-1])


def test_existing_storage_is_refused_on_every_rank(tmp_path):
    """the 'storage already exists' error used to be raised on rank 0 only; the other ranks went on into the
    collectives of create() and hung.  The check is broadcast now: both ranks refuse (and the job ends)."""
    import dist_worker
    from openmmtools_amd.multistate.comm import SingleProcessComm
    dist_worker.run('pt', SingleProcessComm(), n_iter=1, storage_dir=str(tmp_path))          # leaves <tmp>/store behind
    port = 29850 + (os.getpid() % 100)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(HERE, 'dist_worker.py'), 'exists', str(tmp_path)]
    res = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS='1'), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    for r in range(2):
        assert open(os.path.join(tmp_path, 'exists_rank%d.txt' % r)).read() == 'refused'
