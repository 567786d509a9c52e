#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06_14; mkdir -p $O
P="python tools/phase_probe.py"
{
env GO_ITERS=4 GO_PHASES=2 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_CHAIN_TWO=1 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=1 REMD_CHAIN_TWO=1 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_CHAIN_MERGE=0 $P 24 1 seq
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-200 | sed 's/ first .*//' | tee $O/probe.txt
