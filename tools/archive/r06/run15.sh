#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06_15; mkdir -p $O
P="python tools/phase_probe.py"
{
env GO_ITERS=4 GO_PHASES=1 REMD_NB_TUNE_VERBOSE=1 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_NB_TUNE_VERBOSE=1 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_NB_TUNE_VERBOSE=1 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_NB_PRIO=0 REMD_NB_PERSIST_GRID=0 $P 24 1 seq
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-220 | sed 's/ first .*//' | tee $O/probe.txt
