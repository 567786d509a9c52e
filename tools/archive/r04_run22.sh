#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_w
mkdir -p $O
timeout 900 python -m pytest tests/test_compat_groups.py tests/test_distributed_gpu.py tests/test_npt_gpu.py tests/test_work_parity.py tests/test_harmonic_parity.py -m gpu -x -q > $O/pytest_a.log 2>&1; grep -E "passed|failed|rror" $O/pytest_a.log | tail -5
timeout 120 python tools/split_sweep.py auto 24 2>&1 | grep -v amdgpu
