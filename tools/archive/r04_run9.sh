#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_i
mkdir -p $O
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
for rep in 1 2; do
run timeout 120 python tools/split_sweep.py auto 24
for v in pme0 sci3 pme0sci3; do
  run AB_LIB=$(pwd)/openmmtools_amd/libremd_hip_$v.so timeout 120 python tools/split_sweep.py auto 24
done
done
grep -v amdgpu $S
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
