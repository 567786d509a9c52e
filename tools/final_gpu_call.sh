#!/bin/bash
# The end-of-round collection on a GPU box (repo root): the whole GPU suite, smoke, the default bench line, rocprofv3 kernel stats and PMC
# traffic of the same command (tools/collect_profiles.sh), the other BASELINE configurations.
# usage: gpurun --timeout 2700 -- 'bash tools/final_gpu_call.sh r06_final'
tag=${1:-final}
export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
bash tools/collect_profiles.sh $tag > $O/collect.log 2>&1; tail -12 $O/collect.log
for c in 2 4 5; do timeout 500 python tools/bench_configs.py $c 2>> $O/configs.err | cut -c1-400; done | tee $O/bench_configs.jsonl
sha256sum openmmtools_amd/csrc/forces.hip | tee $O/forces_hip.sha256
# the general alchemical path, DHFR on an alchemical ladder, the small NoCutoff systems, the headline ensemble at constant pressure
for c in 4r 4d 5h v g hv n; do timeout 300 python tools/bench_configs.py $c 2>> $O/configs.err | cut -c1-400; done | tee $O/bench_configs_more.jsonl
