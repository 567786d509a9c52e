#!/bin/bash
# round 6, call 5: are the handles' streams sharing hardware queues?  (HIP maps streams onto GPU_MAX_HW_QUEUES = 4 queues by default)
export TMPDIR=/tmp
O=gpurun_out/r06_5; mkdir -p $O
P="python tools/phase_probe.py"
{
for Q in 2 4 8 16; do
GPU_MAX_HW_QUEUES=$Q GO_ITERS=3 $P 24 1 seq
GPU_MAX_HW_QUEUES=$Q GO_ITERS=3 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 REMD_MANY_LEAN=0 $P 24 2 many
GPU_MAX_HW_QUEUES=$Q GO_ITERS=3 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 24 2 many
done
GPU_MAX_HW_QUEUES=8 GO_ITERS=3 $P 24 2 thr
GPU_MAX_HW_QUEUES=8 GO_ITERS=3 $P 24 2 many
GPU_MAX_HW_QUEUES=8 GO_ITERS=3 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 24 3 many
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-200 | tee $O/probe.txt
timeout 300 python -m pytest tests/test_forcefield_parity.py -m gpu -q -x -k "constraint or drift" 2>&1 | tail -5 | tee $O/pytest_constraints.txt
