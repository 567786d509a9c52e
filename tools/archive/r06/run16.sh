#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06_16; mkdir -p $O
P="python tools/phase_probe.py"
{
env GO_ITERS=4 GO_PHASES=2 REMD_CHAIN_PRIO=0 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_CHAIN_PRIO=1 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_CHAIN_PRIO=0 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_CHAIN_PRIO=1 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=1 REMD_CHAIN_PRIO=1 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=1 REMD_CHAIN_PRIO=0 $P 24 1 seq
env GO_ITERS=6 GO_STEPS=100 GO_PHASES=2 REMD_CHAIN_PRIO=0 $P 16 1 seq dhfr
env GO_ITERS=6 GO_STEPS=100 GO_PHASES=2 REMD_CHAIN_PRIO=1 $P 16 1 seq dhfr
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-220 | sed 's/ first .*//' | tee $O/probe.txt
