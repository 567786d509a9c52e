#!/bin/bash
export TMPDIR=/tmp
bash tools/collect_profiles.sh r04_s > /dev/null 2>&1
head -c 600 gpurun_out/r04_s/bench_default.json; echo
head -16 gpurun_out/r04_s/kernel_stats.md | cut -c1-160
cat gpurun_out/r04_s/pmc_traffic.md | cut -c1-170 | head -12
cat gpurun_out/r04_s/kernel_roofs.md | cut -c1-200
python tools/bench_configs.py 2 4 5 5h > gpurun_out/r04_s/bench_configs.jsonl 2> /dev/null; cut -c1-200 gpurun_out/r04_s/bench_configs.jsonl
