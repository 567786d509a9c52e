#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_v
mkdir -p $O
python bench.py --no-cpu-baseline > $O/bench_nocpu.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_nocpu.json')); print(d['value'], d['measured_roofs'])"
python tools/cpu_baseline_full.py > $O/cpu_baseline_full.json 2> /dev/null; cat $O/cpu_baseline_full.json
