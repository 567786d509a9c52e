#!/bin/bash
# round 6, call 7: phases inside one handle (remd_set_phases): bit-identity tests, headline / DHFR / host-guest with one and two blocks, bench line
export TMPDIR=/tmp
O=gpurun_out/r06_7; mkdir -p $O
timeout 900 python -m pytest tests/test_forcefield_parity.py -m gpu -q -x -k "phases or concurrently or stream_modes or resident_pair or constraint" 2>&1 | tail -6 | tee $O/pytest_phases.txt
P="python tools/phase_probe.py"
{
env GO_ITERS=4 GO_PHASES=1 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=2 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=0 $P 24 1 seq
env GO_ITERS=4 GO_PHASES=0 GPU_MAX_HW_QUEUES=4 $P 24 1 seq
env GO_ITERS=3 GO_STEPS=100 GO_PHASES=1 $P 16 1 seq dhfr
env GO_ITERS=3 GO_STEPS=100 GO_PHASES=2 $P 16 1 seq dhfr
env GO_ITERS=3 GO_PHASES=1 $P 8 1 seq hostguest
env GO_ITERS=3 GO_PHASES=2 $P 8 1 seq hostguest
env GO_ITERS=3 GO_PHASES=1 $P 48 1 seq
env GO_ITERS=3 GO_PHASES=2 $P 48 1 seq
env GO_ITERS=3 GO_PHASES=1 $P 16 1 seq
env GO_ITERS=3 GO_PHASES=2 $P 16 1 seq
} 2>&1 | grep -v "amdgpu.ids\|per-replica" | cut -c1-230 | sed 's/ first .*//' | tee $O/probe.txt
timeout 600 python bench.py --no-cpu-baseline --no-shapes > $O/bench_quick.json 2> $O/bench_quick.err; cut -c1-400 $O/bench_quick.json; tail -3 $O/bench_quick.err
