#!/bin/bash
# round 6, call 20: the z passes of power-of-two meshes on register transforms (REMD_PME_POW2 bit 1) beside the plane pass (bit 0)
export TMPDIR=/tmp
O=gpurun_out/r06_20; mkdir -p $O
timeout 900 python -m pytest tests/test_forcefield_parity.py -m gpu -q -x -k "mesh_sizes" 2>&1 | tail -4 | tee $O/pytest_mesh.txt
P="python tools/phase_probe.py"
{
for m in 3 1 0 3 1; do env GO_ITERS=4 GO_PHASES=2 REMD_PME_POW2=$m $P 24 1 seq; done
for m in 3 1 3 1; do env GO_ITERS=6 GO_STEPS=100 GO_PHASES=2 REMD_PME_POW2=$m $P 16 1 seq dhfr; done
for m in 3 1; do env GO_ITERS=3 GO_PHASES=1 REMD_PME_POW2=$m $P 8 1 seq hostguest; done
} 2>&1 | grep -v "amdgpu.ids\|per-replica\|host enqueue" | cut -c1-260 | sed 's/ first .*//' | tee $O/probe.txt
timeout 1200 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_phases_gpu.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
