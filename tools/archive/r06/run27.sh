#!/bin/bash
# round 6, call 27: mesh forces by bin position + one hand-over to the atoms (REMD_PME_FBIN) -- bit-identity, A/B
export TMPDIR=/tmp
O=gpurun_out/r06_27; mkdir -p $O
timeout 900 python -m pytest tests/test_forcefield_parity.py -m gpu -q -x -k "stream_modes or config4 or config5 or large_systems or alch or mesh_sizes" 2>&1 | tail -4 | tee $O/pytest_a.txt
P="python tools/phase_probe.py"
{
for m in 1 0 1 0; do env GO_ITERS=4 GO_PHASES=2 REMD_PME_FBIN=$m $P 24 1 seq; done
for m in 1 0 1 0; do env GO_ITERS=6 GO_STEPS=100 GO_PHASES=2 REMD_PME_FBIN=$m $P 16 1 seq dhfr; done
for m in 1 0; do env GO_ITERS=6 GO_STEPS=100 GO_PHASES=1 REMD_PME_FBIN=$m $P 16 1 seq dhfr; done
for m in 1 0; do env GO_ITERS=3 GO_PHASES=1 REMD_PME_FBIN=$m $P 8 1 seq hostguest; done
} 2>&1 | grep -v "amdgpu.ids\|per-replica\|host enqueue" | cut -c1-260 | sed 's/ first .*//' | tee $O/probe.txt
