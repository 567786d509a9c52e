#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_j
mkdir -p $O
S=$O/sweep.txt; : > $S
run() { env "$@" >> $S 2>&1; }
run REMD_NB_TUNE_VERBOSE=1 timeout 120 python tools/split_sweep.py auto 24
run REMD_NB_PRIO=0 timeout 120 python tools/split_sweep.py auto 24
run REMD_NB_TUNE_VERBOSE=1 timeout 120 python tools/split_sweep.py auto 8 hostguest
run REMD_NB_TUNE_VERBOSE=1 timeout 300 python tools/split_sweep.py auto 16 dhfr
run REMD_NB_TUNE_VERBOSE=1 timeout 120 python tools/split_sweep.py reference 24
grep -v amdgpu $S
python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 300 $O/bench_default.json; echo
timeout 900 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py tests/test_distributed_gpu.py -m gpu -x -q > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
