#!/bin/bash
# round 6, call 3: remd_propagate_many -- two / three handles, one host thread, steps taking turns; lean waits (no fat pollers) or not
export TMPDIR=/tmp
O=gpurun_out/r06_3; mkdir -p $O
P="python tools/phase_probe.py"
{
$P 24 1 seq
$P 24 2 many
REMD_MANY_LEAN=0 $P 24 2 many
REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 24 2 many
REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=256 $P 24 2 many
REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=384 $P 24 2 many
REMD_NB_PRIO=0 REMD_NB_PERSIST_GRID=0 $P 24 2 many
REMD_NB_PRIO=0 REMD_NB_PERSIST_GRID=256 $P 24 2 many
REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 REMD_MANY_LEAN=0 $P 24 2 many
REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 REMD_SYNC_EVENTS=1 $P 24 2 many
REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 24 3 many
REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 24 4 many
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-260 | tee $O/probe.txt
# timeline of the lean two-handle run
ROOT=$(pwd)
(cd /tmp && rm -rf /tmp/tl_many && env GO_STEPS=200 GO_ITERS=2 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 rocprofv3 --kernel-trace -d /tmp/tl_many -o kt -- python $ROOT/tools/phase_probe.py 24 2 many > $O/run_many.txt 2>&1)
python tools/timeline_window.py /tmp/tl_many 500 150 > $O/timeline_many.txt 2>&1; head -3 $O/timeline_many.txt
