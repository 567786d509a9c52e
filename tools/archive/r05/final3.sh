#!/bin/bash
# round 5, final collection on the end-of-session tree: bench line, kernel trace of the same command, PMC traffic, kernel roofs, timeline, the other configs, projection, GPU suite, smoke
export TMPDIR=/tmp
tag=r05_final3
bash tools/collect_profiles.sh $tag > /dev/null 2>&1
head -c 300 gpurun_out/$tag/bench_default.json; echo
head -12 gpurun_out/$tag/kernel_stats.md | cut -c1-160
head -6 gpurun_out/$tag/pmc_traffic.md | cut -c1-170
bash tools/tl_step.sh r05final3 > /dev/null 2>&1; cp gpurun_out/timeline_r05final3.txt gpurun_out/$tag/ 2>/dev/null
python tools/bench_configs.py 2 4 5 5h > gpurun_out/$tag/bench_configs.jsonl 2> /dev/null; cut -c1-170 gpurun_out/$tag/bench_configs.jsonl
python tools/scaling_projection.py alanine dhfr > gpurun_out/$tag/scaling_projection.md 2> /dev/null; tail -30 gpurun_out/$tag/scaling_projection.md | cut -c1-200
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 | tee gpurun_out/$tag/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/$tag/smoke.log
