#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_u
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|rror" $O/pytest_gpu.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
