#!/bin/bash
# round 6, call 6: two hardware queues (GPU_MAX_HW_QUEUES=2) made two interleaved handles FASTER than one (72.9 against 80.6 ms); what
# about three / four groups, the tuner left alone, the handles sharing ONE pair of streams (no environment needed), other systems
export TMPDIR=/tmp
O=gpurun_out/r06_6; mkdir -p $O
P="python tools/phase_probe.py"
F="REMD_MANY_LEAN=0 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0"
{
env GO_ITERS=3 $P 24 1 seq
env GPU_MAX_HW_QUEUES=2 GO_ITERS=3 $F $P 24 2 many
env GPU_MAX_HW_QUEUES=2 GO_ITERS=3 $F $P 24 3 many
env GPU_MAX_HW_QUEUES=2 GO_ITERS=3 $F $P 24 4 many
env GPU_MAX_HW_QUEUES=2 GO_ITERS=4 REMD_MANY_LEAN=0 $P 24 2 many
env GPU_MAX_HW_QUEUES=2 GO_ITERS=3 REMD_MANY_LEAN=0 REMD_NB_PRIO=0 REMD_NB_PERSIST_GRID=0 $P 24 2 many
env GPU_MAX_HW_QUEUES=2 GO_ITERS=3 REMD_MANY_LEAN=0 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=512 $P 24 2 many
env GPU_MAX_HW_QUEUES=2 GO_ITERS=3 REMD_MANY_LEAN=0 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=768 $P 24 2 many
env GPU_MAX_HW_QUEUES=2 GO_ITERS=3 $P 24 2 thr
env GPU_MAX_HW_QUEUES=3 GO_ITERS=3 $F $P 24 2 many
env GPU_MAX_HW_QUEUES=1 GO_ITERS=3 $F $P 24 2 many
env GO_SHARE_STREAMS=1 GO_ITERS=3 $F $P 24 2 many
env GO_SHARE_STREAMS=1 GO_ITERS=3 $F $P 24 3 many
env GO_SHARE_STREAMS=1 GO_ITERS=3 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 24 2 many
env GO_SHARE_STREAMS=1 GPU_MAX_HW_QUEUES=2 GO_ITERS=3 $F $P 24 2 many
env GO_ITERS=3 GO_STEPS=100 $P 16 1 seq dhfr
env GPU_MAX_HW_QUEUES=2 GO_ITERS=3 GO_STEPS=100 $F $P 16 2 many dhfr
env GO_SHARE_STREAMS=1 GO_ITERS=3 GO_STEPS=100 $F $P 16 2 many dhfr
env GO_ITERS=3 $P 8 1 seq hostguest
env GPU_MAX_HW_QUEUES=2 GO_ITERS=3 $F $P 8 2 many hostguest
env GO_SHARE_STREAMS=1 GO_ITERS=3 $F $P 8 2 many hostguest
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-230 | sed 's/ first .*//' | tee $O/probe.txt
