#!/bin/bash
# round 6, call 19: the plane pass of square power-of-two meshes on register transforms (pme_pow2.h) against the scheduled pass
export TMPDIR=/tmp
O=gpurun_out/r06_19; mkdir -p $O
timeout 900 python -m pytest tests/test_forcefield_parity.py -m gpu -q -x -k "mesh_sizes" 2>&1 | tail -4 | tee $O/pytest_mesh.txt
{
python tools/mesh_standalone.py 24
REMD_PME_POW2=0 python tools/mesh_standalone.py 24
python tools/mesh_standalone.py 16 dhfr
REMD_PME_POW2=0 python tools/mesh_standalone.py 16 dhfr
} 2>&1 | grep -v amdgpu.ids | tee $O/standalone.txt
P="python tools/phase_probe.py"
{
env GO_ITERS=4 GO_PHASES=2 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_PME_POW2=0 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 REMD_PME_POW2=0 $P 24 1 seq
env GO_ITERS=6 GO_STEPS=100 GO_PHASES=2 $P 16 1 seq dhfr
env GO_ITERS=6 GO_STEPS=100 GO_PHASES=2 REMD_PME_POW2=0 $P 16 1 seq dhfr
} 2>&1 | grep -v "amdgpu.ids\|per-replica\|host enqueue" | cut -c1-260 | sed 's/ first .*//' | tee $O/probe.txt
timeout 900 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
