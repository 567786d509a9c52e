#!/bin/bash
# round 6, call 18: other raised-priority streams in the process before the engine's (torch's NCCL stream is one): do block B's streams
# still get hardware queues of their own?  (streams_share_a_queue / stream_beside in api.hip)
export TMPDIR=/tmp
O=gpurun_out/r06_18; mkdir -p $O
P="python tools/phase_probe.py"
{
for K in 0 1 2 3; do
env GO_ITERS=3 GO_PHASES=2 REMD_MANY_VERBOSE=1 GO_EXTRA_STREAMS=$K $P 24 1 seq 2>&1 | grep -v "host enqueue"
done
env GO_ITERS=3 GO_PHASES=2 REMD_MANY_VERBOSE=1 GPU_MAX_HW_QUEUES=1 $P 24 1 seq 2>&1 | grep -v "host enqueue"
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-220 | sed 's/ first .*//;s/digest.*//' | tee $O/probe.txt
timeout 600 python -m pytest tests/test_phases_gpu.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
