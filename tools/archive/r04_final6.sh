#!/bin/bash
# the round's last library: GPU suite + smoke (profiles of r04_final5 stand: the non-alchemical kernels are unchanged since)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_final6
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/r04_final6/pytest_gpu.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r04_final6/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_final6/smoke.log 2>&1; tail -1 gpurun_out/r04_final6/smoke.log
