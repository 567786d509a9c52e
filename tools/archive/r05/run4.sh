#!/bin/bash
# round 5, call 4: headline statistics test, bench.py with the extra shapes at N = 1 and as a 2-rank rehearsal on one GPU (gloo), GPU suite
export TMPDIR=/tmp
O=gpurun_out/r05_4; mkdir -p $O
timeout 600 python -m pytest tests/test_headline_statistics_gpu.py -m gpu -x -q -s 2>&1 | tail -12 | tee $O/stats.txt
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_4/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
print(json.dumps(d.get('shapes'), indent=1)[:2500])
print(json.dumps(d.get('cpu_baseline'), indent=1)[:1800])
PY
REMD_BENCH_SHARE_GPU=1 REMD_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; tail -3 $O/bench_n2_gloo.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r05_4/bench_n2_gloo.json').read().strip().splitlines()[-1])
    print('N=2 value', d['value'], 'ms', d['ms_per_step']); print(json.dumps(d.get('shapes'), indent=1)[:2500])
except Exception as e: print('N=2 parse failed', e)
PY
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
