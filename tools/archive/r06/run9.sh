#!/bin/bash
# round 6, call 9: the whole GPU suite on the phased tree + the default bench line
export TMPDIR=/tmp
O=gpurun_out/r06_9; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json; echo
python - <<'PY' | tee $O/shapes.txt
import json
d = json.load(open('gpurun_out/r06_9/bench_default.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', {k: d['roofline'][k] for k in ('achieved', 'frac', 'avg_launch_ms', 'phases')})
print('shapes', json.dumps(d.get('shapes'))[:900])
print('cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
PY
