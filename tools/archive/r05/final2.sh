#!/bin/bash
# the profile part of tools/r05/final.sh again with the profiled command restricted to the headline configuration (--no-shapes)
export TMPDIR=/tmp
bash tools/collect_profiles.sh r05_final2 > /dev/null 2>&1
head -c 300 gpurun_out/r05_final2/bench_default.json; echo
head -14 gpurun_out/r05_final2/kernel_stats.md | cut -c1-170
head -8 gpurun_out/r05_final2/pmc_traffic.md | cut -c1-170
cat gpurun_out/r05_final2/kernel_roofs.md | head -20 | cut -c1-200
bash tools/tl_step.sh r05final > /dev/null 2>&1; head -14 gpurun_out/timeline_r05final.txt; cp gpurun_out/timeline_r05final.txt gpurun_out/r05_final2/
