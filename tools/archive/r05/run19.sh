#!/bin/bash
# round 5, call 19: persistent pair kernel: first item = own index, next ticket drawn ahead of the current item's work
export TMPDIR=/tmp
O=gpurun_out/r05_19; mkdir -p $O
run() { # lib grid R system
  if [ $1 = tree ]; then L=""; else L=$PWD/openmmtools_amd/libremd_hip_$1.so; fi
  if [ $2 = auto ]; then G=""; else G="REMD_NB_PERSIST_GRID=$2"; fi
  env AB_LIB=$L $G REMD_NB_TUNE_VERBOSE=1 python tools/split_sweep.py auto $3 $4 2>&1 | grep "ms per 500\|workgroups" | cut -c1-30,60-230 | sed "s/^/$1 grid=$2 /"
}
for sys in "24 alanine" "8 hostguest"; do for g in auto 640; do for lib in fbase tree; do run $lib $g $sys; done; done; done 2>&1 | tee $O/ab.txt
for lib in fbase tree; do run $lib auto 16 dhfr; done 2>&1 | tee -a $O/ab.txt
timeout 300 python -m pytest tests/test_forcefield_parity.py -m gpu -q -p no:cacheprovider -x -k "resident or bit or alanine" 2>&1 | tail -2 | tee -a $O/ab.txt
