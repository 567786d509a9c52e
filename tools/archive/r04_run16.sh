#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_q
mkdir -p $O
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
run REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 REMD_NB_FOLD=0 timeout 120 python tools/split_sweep.py auto 24
run REMD_NB_PRIO=1 REMD_NB_PERSIST_GRID=0 timeout 120 python tools/split_sweep.py auto 24
run REMD_NB_TUNE_VERBOSE=1 REMD_NB_FOLD=0 timeout 120 python tools/split_sweep.py auto 24
run REMD_NB_TUNE_VERBOSE=1 timeout 120 python tools/split_sweep.py auto 24
grep -v amdgpu $S
