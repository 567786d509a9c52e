"""Worker of tests/test_comm_capi.py: one rank of a run sharded through the library's own RCCL communicator (no torch.distributed;
the 128-byte communicator id travels through a file, standing in for the MPI_Bcast of a compiled host).

    python tests/comm_worker_rccl.py <rank> <world> <outdir>
Each rank: its block of 6 LJ-fluid replicas on GPU <rank>, 3 iterations of propagate -> u_kl -> all-gather -> swap-all on the
handle's own matrix; writes labels / acceptance counts / its rows per iteration to <outdir>/rank<r>.npz."""
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

KB = 0.008314462618153242
R_GLOBAL, N_ITER = 6, 3


def main(rank, world, outdir):
    from openmmtools_amd import testsystems as ts
    from openmmtools_amd.system import system_to_desc
    from openmmtools_amd._engine import HipEngine
    from openmmtools_amd.multistate.comm import block_partition
    lj = ts.LennardJonesFluid(nparticles=216)
    box = np.diag(lj.system.getDefaultPeriodicBoxVectors())
    eng = HipEngine(device=rank)
    eng.set_system(system_to_desc(lj.system))
    eng.set_states(1.0 / (KB * np.linspace(100.0, 150.0, R_GLOBAL)))
    eng.set_integrator('V R O R V', 0.001, 1.0, 25, True, 1e-8)
    eng.seed(2024)
    begins, counts = block_partition(R_GLOBAL, world)
    b, c = begins[rank], counts[rank]
    labels = np.arange(R_GLOBAL)
    eng.set_replicas(R_GLOBAL, b, np.tile(lj.positions, (c, 1, 1)), None, np.tile(box, (c, 1)), labels)
    if world > 1 or os.environ.get('COMM_WORLD_OF_ONE'):
        idfile = os.path.join(outdir, 'comm_id')
        if rank == 0:
            with open(idfile + '.tmp', 'wb') as fh:
                fh.write(eng.comm_unique_id())
            os.replace(idfile + '.tmp', idfile)
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise RuntimeError('no communicator id from rank 0')
            time.sleep(0.05)
        with open(idfile, 'rb') as fh:
            eng.comm_init(rank, world, fh.read())
    out = {}
    for it in range(N_ITER):
        assert not eng.propagate(it).any()
        rows = eng.compute_energies()                       # local rows, also written into the handle's own matrix
        eng.comm_all_gather_energies()
        labels, nacc, nprop, _ = eng.mix('swap-all', it, labels)
        eng.set_labels(labels)
        out['labels%d' % it], out['nacc%d' % it], out['nprop%d' % it], out['rows%d' % it] = labels, nacc, nprop, rows
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), begin=b, count=c, **out)
    eng.close()


if __name__ == '__main__':
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
