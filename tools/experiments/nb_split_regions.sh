export TMPDIR=/tmp
run() { echo -n "$1 config $2: "; env $1 timeout 300 python tools/bench_configs.py $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f it/s' % d['iterations_per_s'])"; }
for c in 4r 4d 4 5 5h; do run X=0 $c; done
