# swap-all on a high-acceptance ensemble (22-atom molecule, 300 - 600 K): speculative windows + whole-chip preparation (REMD_MIX_FLOW=0)
# against the dataflow kernel (1), by ensemble size
export TMPDIR=/tmp; O=gpurun_out/r06s3_27; mkdir -p $O
for nt in 24 64 128; do for flow in 0 1; do
  REMD_BENCH_NT=$nt REMD_MIX_FLOW=$flow timeout 300 python tools/bench_configs.py v 2>>$O/err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('nt $nt flow $flow  mixing %.3f ms  iteration %.2f ms' % (1e3 * d['timing']['mixing_seconds'], d['ms_per_iteration']))"
done; done | tee $O/summary.txt
for flow in 0 1; do REMD_MIX_FLOW=$flow timeout 300 python tools/bench_configs.py 2 2>>$O/err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('config 2 (16 lambda states) flow $flow  mixing %.3f ms  iteration %.2f ms' % (1e3 * d['timing']['mixing_seconds'], d['ms_per_iteration']))"
done | tee -a $O/summary.txt
