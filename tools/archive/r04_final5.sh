#!/bin/bash
# the profile part of tools/r04_final4.sh again (that call landed on a box that ran everything 10 % slower: 92.9 ms per iteration,
# against 83.7 for the same tree on the boxes of the calls before and after); the test logs of r04_final4 stand
export TMPDIR=/tmp
bash tools/collect_profiles.sh r04_final5 > /dev/null 2>&1
head -c 300 gpurun_out/r04_final5/bench_default.json; echo
python tools/bench_configs.py 2 4 5 5h > gpurun_out/r04_final5/bench_configs.jsonl 2> /dev/null; cut -c1-160 gpurun_out/r04_final5/bench_configs.jsonl
