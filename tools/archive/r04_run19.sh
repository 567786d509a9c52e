#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_t
mkdir -p $O
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
for rep in 1 2; do
run REMD_LISTED_RIDE=0 timeout 120 python tools/split_sweep.py auto 24
run timeout 120 python tools/split_sweep.py auto 24
done
run timeout 120 python tools/split_sweep.py auto 8 hostguest
run REMD_LISTED_RIDE=0 timeout 120 python tools/split_sweep.py auto 8 hostguest
run timeout 300 python tools/split_sweep.py auto 16 dhfr
run REMD_LISTED_RIDE=0 timeout 300 python tools/split_sweep.py auto 16 dhfr
grep -v amdgpu $S
timeout 900 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_distributed_gpu.py tests/test_harmonic_parity.py tests/test_mts_parity.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | grep -v ROCm
