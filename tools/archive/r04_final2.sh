#!/bin/bash
export TMPDIR=/tmp
bash tools/collect_profiles.sh r04_final2 > /dev/null 2>&1
head -c 300 gpurun_out/r04_final2/bench_default.json; echo
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r04_final2/pytest_gpu.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r04_final2/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_final2/smoke.log 2>&1; tail -1 gpurun_out/r04_final2/smoke.log
python tools/bench_configs.py 2 4 5 5h > gpurun_out/r04_final2/bench_configs.jsonl 2> /dev/null; cut -c1-160 gpurun_out/r04_final2/bench_configs.jsonl
