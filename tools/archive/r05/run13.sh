#!/bin/bash
# round 5, call 13: swap-all, speculative windows: both chains walked at once + walks skipped when none of the words read last time changed
export TMPDIR=/tmp
O=gpurun_out/r05_13; mkdir -p $O
timeout 900 python -m pytest tests/test_mix_parity.py tests/test_reference_golden.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/parity.txt
for lib in mixbase tree mixbase tree; do
  if [ $lib = tree ]; then L=""; else L=$PWD/openmmtools_amd/libremd_hip_$lib.so; fi
  echo "== $lib pt"; AB_LIB=$L REMD_MIX_FLOW=0 timeout 300 python tools/mix_microbench.py 24 64 128 192 2>&1 | grep "^R "
done 2>&1 | tee $O/mix.txt
for lib in mixbase tree; do
  if [ $lib = tree ]; then L=""; else L=$PWD/openmmtools_amd/libremd_hip_$lib.so; fi
  for fl in 0 1; do echo "== $lib hot FLOW=$fl"; AB_LIB=$L MIX_MATRIX=hot REMD_MIX_FLOW=$fl timeout 300 python tools/mix_microbench.py 24 128 192 2>&1 | grep "^R "; done
done 2>&1 | tee -a $O/mix.txt
echo "== tree pt, debug"; REMD_MIX_FLOW=0 REMD_MIX_DEBUG=1 timeout 300 python tools/mix_microbench.py 128 192 2>&1 | grep "mix-pre" | awk 'NR%7==1' | tee -a $O/mix.txt
for pr in 4 5 8; do echo "== tree pt PERR=$pr"; REMD_MIX_PERR=$pr REMD_MIX_FLOW=0 timeout 300 python tools/mix_microbench.py 64 128 192 2>&1 | grep "^R "; done 2>&1 | tee -a $O/mix.txt
