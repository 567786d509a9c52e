#!/bin/bash
# round 6, call 4: what a launch costs the host on this stack; where remd_propagate_many's time goes; constraint-tolerance test + chain A/B
export TMPDIR=/tmp
O=gpurun_out/r06_4; mkdir -p $O
./tools/probes/launch_rate 2>&1 | tee $O/launch_rate.txt
P="python tools/phase_probe.py"
{
REMD_MANY_VERBOSE=1 GO_ITERS=3 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 $P 24 2 many
REMD_MANY_VERBOSE=1 GO_ITERS=3 REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 REMD_MANY_LEAN=0 $P 24 2 many
GO_ITERS=5 $P 24 1 seq
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-260 | tee $O/probe.txt
timeout 600 python -m pytest tests/test_forcefield_parity.py -m gpu -q -x -k "constraint or drift or substeps or stream_modes or resident_pair" 2>&1 | tail -5 | tee $O/pytest_constraints.txt
timeout 300 python -m pytest tests/test_integrator_program.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest_program.txt
