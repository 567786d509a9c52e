#!/bin/bash
# scaling projection + the N > 1 bench flow on one GPU (gloo, shared device) + the round's profile set
export TMPDIR=/tmp
O=gpurun_out/r04_m
mkdir -p $O
timeout 1200 python tools/scaling_projection.py alanine dhfr > $O/scaling_projection.md 2> $O/scaling_projection.err; cat $O/scaling_projection.md
for N in 2 4; do
  REMD_BENCH_SHARE_GPU=1 REMD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N \
     bench.py --gpus $N --steps 2 --warmup 1 > $O/bench_gloo_shared_gpu_N$N.json 2> $O/bench_gloo_shared_gpu_N$N.err
  echo "N=$N rc=$?"; head -c 500 $O/bench_gloo_shared_gpu_N$N.json; echo
done
REMD_BENCH_SHARE_GPU=1 REMD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
     bench.py --gpus 2 --steps 2 --warmup 1 --replicas-total 24 > $O/bench_gloo_shared_gpu_N2_strong24.json 2> /dev/null; head -c 300 $O/bench_gloo_shared_gpu_N2_strong24.json; echo
