#!/bin/bash
# round 6, call 1: (a) why do split handles give other bits than one handle (sequential vs threads, tuner pinned or not, events or flags);
# (b) what two 12-replica handles cost without resident pollers; (c) the integrator-program check of round 5 that never ran
export TMPDIR=/tmp
O=gpurun_out/r06_1; mkdir -p $O
P="python tools/phase_probe.py"
{
$P 24 1 seq
$P 24 2 seq
$P 24 2 thr
REMD_NB_PERSIST_GRID=0 $P 24 1 seq
REMD_NB_PERSIST_GRID=0 $P 24 2 seq
REMD_NB_PERSIST_GRID=0 $P 24 2 thr
REMD_SYNC_EVENTS=1 REMD_NB_PERSIST_GRID=0 $P 24 1 seq
REMD_SYNC_EVENTS=1 REMD_NB_PERSIST_GRID=0 $P 24 2 thr
REMD_SYNC_EVENTS=1 REMD_NB_PERSIST_GRID=256 $P 24 2 thr
REMD_SYNC_EVENTS=1 REMD_NB_PERSIST_GRID=384 $P 24 2 thr
REMD_SYNC_EVENTS=1 REMD_NB_PERSIST_GRID=0 $P 24 3 thr
REMD_SYNC_EVENTS=1 REMD_NB_PERSIST_GRID=0 $P 12 1 seq
$P 12 1 seq
} 2>&1 | grep -v "amdgpu.ids" | tee $O/probe.txt
timeout 300 python -m pytest tests/test_integrator_program.py -m gpu -q 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tee $O/integrator_program.txt
