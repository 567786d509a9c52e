#!/bin/bash
# round 6, call 34: the 128 x 128 plane pass compiled for 96 registers (five wavefronts per SIMD) so that it fits beside two pair workgroups,
# against the 122-register compilation (libremd_hip_xy4.so); the laundered inverse-half addresses (no spills) are in both
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_34; mkdir -p $O
P="python tools/phase_probe.py"
{
for lib in "" xy4 "" xy4; do
  L=""; [ -n "$lib" ] && L="AB_LIB=$ROOT/openmmtools_amd/libremd_hip_$lib.so"
  echo "lib=[$lib]"
  env GO_ITERS=6 GO_STEPS=100 GO_PHASES=2 REMD_NB_TUNE_VERBOSE=1 $L $P 16 1 seq dhfr
done
for lib in "" xy4; do
  L=""; [ -n "$lib" ] && L="AB_LIB=$ROOT/openmmtools_amd/libremd_hip_$lib.so"
  echo "lib=[$lib]"
  env GO_ITERS=6 GO_STEPS=100 GO_PHASES=1 $L $P 16 1 seq dhfr
  env GO_ITERS=4 GO_PHASES=2 $L $P 24 1 seq
done
} 2>&1 | grep -v "amdgpu.ids\|per-replica\|host enqueue" | cut -c1-300 | sed 's/ first .*//' | tee $O/probe.txt
timeout 900 python -m pytest tests/test_forcefield_parity.py -m gpu -q -x -k "mesh_sizes or register_transforms or config5 or large" 2>&1 | tail -3 | tee $O/pytest.txt
