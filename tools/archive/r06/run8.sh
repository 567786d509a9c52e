#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06_8; mkdir -p $O
timeout 900 python -m pytest tests/test_forcefield_parity.py -m gpu -q -x -k "phases or concurrently" 2>&1 | tail -6 | tee $O/pytest_phases.txt
P="python tools/phase_probe.py"
{
env GO_ITERS=8 GO_STEPS=100 GO_PHASES=1 $P 16 1 seq dhfr
env GO_ITERS=8 GO_STEPS=100 GO_PHASES=2 $P 16 1 seq dhfr
env GO_ITERS=4 GO_STEPS=100 GO_PHASES=2 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 16 1 seq dhfr
env GO_ITERS=4 GO_STEPS=100 GO_PHASES=1 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 16 1 seq dhfr
env GO_ITERS=4 GO_PHASES=2 REMD_MANY_LEAN=1 $P 24 1 seq
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-260 | sed 's/ first .*//' | tee $O/probe.txt
