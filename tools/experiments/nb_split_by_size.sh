export TMPDIR=/tmp
run() { echo -n "$1 config $2: "; env $1 timeout 300 python tools/bench_configs.py $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f it/s' % d['iterations_per_s'])"; }
for rep in 1 2; do run X=0 5; run REMD_NB_SPLIT=4 5; done
for rep in 1 2; do run X=0 4; run REMD_NB_SPLIT=4 4; done
run X=0 5h; run REMD_NB_SPLIT=4 5h
