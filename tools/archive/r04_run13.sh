#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_n
mkdir -p $O
for pr in 2 3 4 5 6; do
  echo "== REMD_MIX_PERR=$pr" >> $O/mix.txt
  REMD_MIX_PERR=$pr REMD_MIX_FLOW=0 timeout 300 python tools/mix_microbench.py 24 64 128 192 2>&1 | grep "^R " >> $O/mix.txt
done
echo "== hot matrix, by kernel" >> $O/mix.txt
for fl in 0 1; do for pr in 3 6; do echo "-- FLOW=$fl PERR=$pr" >> $O/mix.txt; MIX_MATRIX=hot REMD_MIX_PERR=$pr REMD_MIX_FLOW=$fl timeout 300 python tools/mix_microbench.py 24 128 192 2>&1 | grep "^R " >> $O/mix.txt; done; done
cat $O/mix.txt
