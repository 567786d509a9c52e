#!/bin/bash
# round 5, final collection: bench line, kernel trace of the same command, PMC traffic, kernel roofs, the other configs, scaling projection, GPU suite
export TMPDIR=/tmp
tag=${1:-r05_final}
bash tools/collect_profiles.sh $tag > /dev/null 2>&1
head -c 400 gpurun_out/$tag/bench_default.json; echo
head -12 gpurun_out/$tag/kernel_stats.md | cut -c1-160
python tools/bench_configs.py 2 4 5 5h > gpurun_out/$tag/bench_configs.jsonl 2> /dev/null; cut -c1-170 gpurun_out/$tag/bench_configs.jsonl
python tools/scaling_projection.py alanine dhfr > gpurun_out/$tag/scaling_projection.md 2> /dev/null; head -50 gpurun_out/$tag/scaling_projection.md
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 | tee gpurun_out/$tag/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/$tag/smoke.log
