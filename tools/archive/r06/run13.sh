#!/bin/bash
# round 6, call 13: queue priorities of the two blocks' streams (two handles, steps taking turns, 2 hardware queues per priority)
export TMPDIR=/tmp
O=gpurun_out/r06_13; mkdir -p $O
P="python tools/phase_probe.py"
{
env GO_ITERS=4 $P 24 2 many
env GO_ITERS=4 REMD_MAIN_PRIO=1 REMD_DIRECT_PRIO=0 $P 24 2 many
env GO_ITERS=4 REMD_DIRECT_PRIO=0 $P 24 2 many
env GO_ITERS=4 REMD_MAIN_PRIO=1 $P 24 2 many
env GO_ITERS=4 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 24 2 many
env GO_ITERS=4 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 REMD_MAIN_PRIO=1 REMD_DIRECT_PRIO=0 $P 24 2 many
env GO_ITERS=4 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 REMD_DIRECT_PRIO=0 $P 24 2 many
env GO_ITERS=4 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 REMD_MAIN_PRIO=1 $P 24 2 many
env GO_ITERS=4 REMD_NB_PRIO=0 REMD_NB_PERSIST_GRID=0 REMD_MAIN_PRIO=1 REMD_DIRECT_PRIO=0 $P 24 2 many
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-260 | sed 's/ first .*//' | tee $O/probe.txt
