export TMPDIR=/tmp
run() { echo -n "$1: "; env $1 timeout 300 python tools/bench_configs.py 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f it/s' % d['iterations_per_s'])"; }
for k in X=0 REMD_PME_CHAINBIN=1 REMD_PME_CHAINBIN=0 REMD_NB_RANK=0 REMD_PHASES=1 REMD_NB_FOLD=0 REMD_LISTED_RIDE=0 REMD_LISTED_MAIN=0 REMD_NB_RESORT=80 REMD_NB_HBITS=6 X=1; do run "$k"; done
