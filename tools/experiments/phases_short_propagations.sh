export TMPDIR=/tmp
run() { echo -n "$1 md-steps $2: "; env $1 python bench.py --no-cpu-baseline --no-shapes --steps 20 --warmup 3 --md-steps $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms per iteration, propagation %.3f ms' % (d['ms_per_step'], 1e3*d['timing']['propagation_seconds']))"; }
for n in 1 5 20 50; do run REMD_PHASES=1 $n; run X=0 $n; done
