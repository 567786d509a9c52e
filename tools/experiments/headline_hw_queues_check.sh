export TMPDIR=/tmp
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-shapes --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f it/s  %.2f ms  phases %s' % (d['value'], d['ms_per_step'], d['roofline'].get('phases')))"; }
for k in X=0 GPU_MAX_HW_QUEUES=1 "GPU_MAX_HW_QUEUES=3 REMD_PHASES=2" X=1; do run "$k"; done
