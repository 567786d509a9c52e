"""The sharding collectives behind the C ABI (include/remd_hip.h: remd_comm_unique_id / _init / _all_gather_energies / _finalize;
csrc/comm.hip): what a host without torch.distributed binds to run one process per GPU.  Reference seam: mpiplus distributes
the replicas and gathers their energies (multistatesampler.py:1296-1311, 1448-1449), replicaexchange.py:255 mixes."""
import os
import subprocess
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, 'comm_worker_rccl.py')
KB = 0.008314462618153242


def _run_ranks(world, outdir, env=None):
    e = dict(os.environ, **(env or {}))
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(outdir)], env=e, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0].decode())
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append('timeout')
    assert all(p.returncode == 0 for p in procs), '\n'.join(logs)[-4000:]
    return [np.load(os.path.join(str(outdir), 'rank%d.npz' % r)) for r in range(world)]


def test_cpu_library_is_a_world_of_one():
    import oracle
    from openmmtools_amd._engine import HipEngine
    lib = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')
    if not os.path.exists(lib):
        oracle.build()
    eng = HipEngine(lib_path=lib)
    uid = eng.comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(0, 1, uid)
    with pytest.raises(RuntimeError, match='not implemented in the CPU library'):
        eng.comm_init(0, 2, uid)
    with pytest.raises(RuntimeError):
        eng.comm_init(1, 1, uid)
    eng.comm_finalize()


def _lj_engine(eng, R_global, begin, count):
    from openmmtools_amd import testsystems as ts
    from openmmtools_amd.system import system_to_desc
    lj = ts.LennardJonesFluid(nparticles=216)
    box = np.diag(lj.system.getDefaultPeriodicBoxVectors())
    eng.set_system(system_to_desc(lj.system))
    eng.set_states(1.0 / (KB * np.linspace(100.0, 150.0, R_global)))
    eng.set_integrator('V R O R V', 0.001, 1.0, 5, True, 1e-8)
    eng.seed(3)
    eng.set_replicas(R_global, begin, np.tile(lj.positions, (count, 1, 1)), None, np.tile(box, (count, 1)), np.arange(R_global))


@pytest.mark.gpu
def test_world_of_one_on_rccl_and_the_refusals(hip_engine_factory):
    """RCCL itself at world 1 (communicator, block exchange, grouped in-place broadcasts on the handle's stream); a sharded
    handle without a communicator, and blocks that do not cover the replicas, are refused with the reason."""
    eng = hip_engine_factory()
    _lj_engine(eng, 4, 0, 4)
    eng.comm_all_gather_energies()                                   # unsharded, no communicator: nothing to do
    uid = eng.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(0, 1, uid)
    rows = eng.compute_energies()
    eng.comm_all_gather_energies()
    got = eng.mix('swap-all', 0, np.arange(4))
    want = eng.mix_host('swap-all', 0, rows, np.arange(4))
    assert all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3]))
    _lj_engine(eng, 4, 0, 2)                                         # half the replicas on a world of one
    eng.compute_energies()
    with pytest.raises(RuntimeError, match='hold 2 of 4 replicas'):
        eng.comm_all_gather_energies()
    eng.comm_finalize()
    with pytest.raises(RuntimeError, match='remd_comm_init was not called'):
        eng.comm_all_gather_energies()
    with pytest.raises(RuntimeError, match='rank 2 of 2'):
        eng.comm_init(2, 2, uid)


@pytest.mark.gpu
def test_single_rank_worker_follows_the_same_run_with_and_without_the_communicator(tmp_path):
    for d in ('plain', 'rccl'):
        (tmp_path / d).mkdir()
    a = _run_ranks(1, tmp_path / 'plain')[0]
    b = _run_ranks(1, tmp_path / 'rccl', env={'COMM_WORLD_OF_ONE': '1'})[0]
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2 or not os.environ.get('REMD_TEST_MULTI_GPU'),
                    reason='needs two GPUs (RCCL refuses two ranks on one device) and REMD_TEST_MULTI_GPU=1: this path has not run on hardware yet')
def test_two_ranks_over_xgmi_follow_the_single_process_run(tmp_path):
    """Blocks of 3 + 3 replicas on two GPUs, rows exchanged by the library's RCCL all-gather: labels, acceptance counts and
    every rank's rows equal the single-process run bit for bit (noise is keyed by the global replica index)."""
    for d in ('one', 'two'):
        (tmp_path / d).mkdir()
    one = _run_ranks(1, tmp_path / 'one')[0]
    two = _run_ranks(2, tmp_path / 'two')
    for it in range(3):
        for r in two:
            assert np.array_equal(r['labels%d' % it], one['labels%d' % it])
            assert np.array_equal(r['nacc%d' % it], one['nacc%d' % it]) and np.array_equal(r['nprop%d' % it], one['nprop%d' % it])
            b, c = int(r['begin']), int(r['count'])
            assert np.array_equal(r['rows%d' % it], one['rows%d' % it][b:b + c])
