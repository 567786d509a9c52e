export TMPDIR=/tmp
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-shapes --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f it/s  %.2f ms' % (d['value'], d['ms_per_step']))"; }
for k in X=0 REMD_PAIR_AFTER_XY=1 REMD_PME_RADIX8=0 "REMD_PME_POW2=1" "REMD_PME_POW2=2" X=1; do run "$k"; done
