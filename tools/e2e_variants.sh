#!/bin/bash
# usage (GPU box, repo root): tools/e2e_variants.sh "ENV=1 ..." "ENV2=..."   -> short bench.py run (3 iterations) per environment variant
for v in "$@"; do
  env $v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],3), 'it/s', round(d['ms_per_step'],2), 'ms')"
done
