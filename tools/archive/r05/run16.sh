#!/bin/bash
# round 5, call 16: chain: program tokens from registers + every force-independent load in front of the wait for the forces
export TMPDIR=/tmp
O=gpurun_out/r05_16; mkdir -p $O
echo "== stamps (tree)"; AB_LIB=$PWD/openmmtools_amd/libremd_hip_stamps.so python tools/chain_segments.py 24 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tee $O/segments.txt
python tools/ab_libs.py --R 24 --system alanine --rounds 3 head late tree 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tee $O/ab.txt
python tools/ab_libs.py --R 8 --system hostguest --rounds 2 head late tree 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tee -a $O/ab.txt
for lib in head tree head tree; do if [ $lib = tree ]; then L=""; else L=$PWD/openmmtools_amd/libremd_hip_$lib.so; fi; AB_LIB=$L python tools/split_sweep.py auto 16 dhfr 2>&1 | tail -1 | cut -c60-220; done | tee -a $O/ab.txt
timeout 600 python -m pytest tests/test_forcefield_parity.py tests/test_harmonic_parity.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2 | tee -a $O/ab.txt
