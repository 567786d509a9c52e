#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_o
mkdir -p $O
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
run timeout 120 python tools/split_sweep.py auto 24
for sp in 4 12 16; do run REMD_NB_SPLIT=$sp timeout 120 python tools/split_sweep.py auto 24; done
for rs in 20 80; do run REMD_NB_RESORT=$rs timeout 120 python tools/split_sweep.py auto 24; done
run timeout 120 python tools/split_sweep.py auto 24
grep -v amdgpu $S
