export TMPDIR=/tmp
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-shapes --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f it/s  %.2f ms  launches timed %s' % (d['value'], d['ms_per_step'], d['roofline'].get('launches')))"; }
for k in X=0 REMD_PROF_EVERY=64 REMD_PROF_EVERY=256 REMD_PROF_EVERY=4 X=1; do run "$k"; done
