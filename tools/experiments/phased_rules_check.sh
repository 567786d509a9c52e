export TMPDIR=/tmp; O=gpurun_out/r06s3_37; mkdir -p $O
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-shapes --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f it/s  %.2f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do for k in X=0 "REMD_PME_CHAINBIN=1 REMD_NB_RANK=0" X=1; do run "$k"; done; done 2>&1 | tee $O/summary.txt
timeout 900 python -m pytest tests/test_phases_gpu.py tests/test_forcefield_parity.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -6 | tee $O/pytest.txt
for c in 4 5; do timeout 300 python tools/bench_configs.py $c 2>/dev/null | cut -c1-200; done | tee -a $O/summary.txt
