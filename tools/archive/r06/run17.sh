#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06_17; mkdir -p $O
P="python tools/phase_probe.py"
{
for S in 8 12 16 4; do
env GO_ITERS=4 GO_PHASES=2 REMD_NB_SPLIT=$S $P 24 1 seq
done
env GO_ITERS=4 GO_PHASES=2 REMD_NB_RESORT=80 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_NB_RESORT=20 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_PME_XYT=256 $P 24 1 seq
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-220 | sed 's/ first .*//;s/digest.*//' | tee $O/probe.txt
