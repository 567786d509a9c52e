#!/bin/bash
# round 6, call 30: DHFR digests and times with the four-wavefront list builder, its one-wavefront compilation, and the old builder
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06_30; mkdir -p $O
P="python tools/phase_probe.py"
{
for lib in "" list1 listold "" listold; do
  L=""; [ -n "$lib" ] && L="AB_LIB=$ROOT/openmmtools_amd/libremd_hip_$lib.so"
  echo "lib=[$lib]"
  env GO_ITERS=5 GO_STEPS=100 GO_PHASES=2 REMD_MANY_VERBOSE=1 REMD_NB_TUNE_VERBOSE=1 $L $P 16 1 seq dhfr
done
} 2>&1 | grep -v "amdgpu.ids\|per-replica\|host enqueue" | cut -c1-300 | sed 's/ first .*//' | tee $O/probe.txt
