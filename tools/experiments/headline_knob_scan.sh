# end of round 6: the run-time switches that bear on the headline step, re-scanned on the final tree in one call (same box, interleaved base lines)
export TMPDIR=/tmp; O=gpurun_out/r06s3_35; mkdir -p $O
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-shapes --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f it/s  %.2f ms' % (d['value'], d['ms_per_step']))"; }
for k in X=0 REMD_NB_RESORT=20 REMD_NB_RESORT=80 X=1 REMD_NB_PRIO=0 REMD_NB_PRIO=1 REMD_NB_RANK=0 REMD_NB_RANK=1 X=2 REMD_NB_FOLD=0 REMD_LISTED_RIDE=0 REMD_LISTED_MAIN=0 REMD_NB_HBITS=4 REMD_NB_HBITS=6 X=3 REMD_CHAIN_TWO=1 REMD_PME_CHAINBIN=0 REMD_CHAIN_MERGE=0 X=4; do run $k; done 2>&1 | tee $O/summary.txt
