#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_x
mkdir -p $O
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
for rep in 1 2; do
run AB_LIB=$(pwd)/openmmtools_amd/libremd_hip_nointc.so timeout 120 python tools/split_sweep.py auto 24
run timeout 120 python tools/split_sweep.py auto 24
done
run AB_LIB=$(pwd)/openmmtools_amd/libremd_hip_nointc.so timeout 120 python tools/split_sweep.py auto 24 alanine standalone
run timeout 120 python tools/split_sweep.py auto 24 alanine standalone
run AB_LIB=$(pwd)/openmmtools_amd/libremd_hip_nointc.so timeout 300 python tools/split_sweep.py auto 16 dhfr
run timeout 300 python tools/split_sweep.py auto 16 dhfr
grep -v amdgpu $S
timeout 900 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_distributed_gpu.py tests/test_npt_gpu.py -m gpu -x -q > $O/pytest_a.log 2>&1; grep -E "passed|failed|rror" $O/pytest_a.log | tail -3
