#!/bin/bash
# round 5, call 3: why are the lists one step behind slower?  kernel timelines of both modes + the rest of the GPU suite
export TMPDIR=/tmp
O=gpurun_out/r05_3; mkdir -p $O
bash tools/tl_step.sh pipe > /dev/null 2>&1; cp gpurun_out/timeline_pipe.txt $O/
bash tools/tl_step.sh exact REMD_LIST_PIPE=0 > /dev/null 2>&1; cp gpurun_out/timeline_exact.txt $O/
head -40 $O/timeline_pipe.txt; head -32 $O/timeline_exact.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
