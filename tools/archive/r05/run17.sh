#!/bin/bash
# round 5, call 17: chain2 (DHFR) with late loads + tokens from registers against HEAD; parity
export TMPDIR=/tmp
O=gpurun_out/r05_17; mkdir -p $O
for lib in head tree head tree; do if [ $lib = tree ]; then L=""; else L=$PWD/openmmtools_amd/libremd_hip_$lib.so; fi; AB_LIB=$L python tools/split_sweep.py auto 16 dhfr 2>&1 | tail -1 | cut -c60-220; done | tee $O/ab.txt
python tools/ab_libs.py --R 24 --system alanine --rounds 2 head tree 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tee -a $O/ab.txt
timeout 600 python -m pytest tests/test_forcefield_parity.py tests/test_harmonic_parity.py tests/test_work_parity.py tests/test_mts_parity.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2 | tee -a $O/ab.txt
