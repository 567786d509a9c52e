#!/bin/bash
# usage: tools/prof_variants.sh "<ENV=..>" ["<ENV=..>" ...]   (run on the GPU box, from the repo root)
# For every environment variant: rocprofv3 kernel trace of 2 x 200 MD steps of the 24-replica alanine system with the
# stream overlap switched off (standalone kernel durations), reduced to a short per-kernel table.
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/variants
n=0
for v in "$@"; do
  n=$((n+1))
  out=/tmp/prof_$n
  rm -rf $out
  (cd /tmp && env REMD_OVERLAP=0 $v rocprofv3 --kernel-trace -d $out -o v -- python $ROOT/tools/small_r_profile.py ${NREP:-24} > /dev/null 2>&1)
  db=$(find $out -name "*.db" | head -1)
  echo "=== variant $n: $v"
  python tools/rocpd_stats.py $db gpurun_out/variants/v$n.md | awk -F'|' 'NR>2 && NF>5 {printf "%-60s calls %6s avg_us %8s\n", $2, $3, $5}' | head -${TOPN:-14}
done
