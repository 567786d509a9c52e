#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_b
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
REMD_TOOLS_EWALD_SPLIT=auto bash tools/pmc_step.sh $O/pmc_step_auto.md > /dev/null 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 400 $O/bench_default.json; echo
cat $O/pmc_step_auto.md | cut -c1-220
