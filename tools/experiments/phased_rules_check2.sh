export TMPDIR=/tmp; O=gpurun_out/r06s3_39; mkdir -p $O
python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline %.3f it/s' % d['value'], 'traffic_is_this_kernel', d['roofline'].get('traffic_source_is_this_kernel'))
for k,v in d['shapes'].items(): print(k, v.get('value'), v.get('ms_per_iteration'))" | tee $O/summary.txt
for k in X=0 REMD_PHASES=2; do echo -n "$k config 4: "; env $k timeout 300 python tools/bench_configs.py 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f it/s' % d['iterations_per_s'])"; done | tee -a $O/summary.txt
for R in 8 12; do for k in X=0 REMD_PHASES=2; do echo -n "$k alanine R=$R: "; env $k python bench.py --no-cpu-baseline --no-shapes --steps 10 --warmup 2 --replicas-total $R 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f units/s  %.2f ms' % (d['value'], d['ms_per_step']))"; done; done | tee -a $O/summary.txt
