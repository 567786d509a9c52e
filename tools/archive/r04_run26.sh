#!/bin/bash
# DHFR step timeline (16 replicas)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_z
cd /tmp && DHFR_STEPS=100 DHFR_ITERS=4 rocprofv3 --kernel-trace -d /tmp/tl -o dhfr -- python $GRAFT_REPO_ROOT/tools/dhfr_profile.py dhfr 16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/r04_z/dhfr_timeline.txt
import glob, sqlite3
db = sqlite3.connect(glob.glob('/tmp/tl/**/*.db', recursive=True)[0])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'integrate_chain_kernel' in r[0]]
k = len(idx) - 30
a, b = idx[k], idx[k + 2]
t0 = rows[a][1]
qs = sorted(set(r[3] for r in rows[a:b + 1]))
print("wall us per step", (rows[b][1] - rows[a][1]) / 1e3 / 2)
for r in rows[a:b + 1]:
    print("  %-44s q%d start %8.1f dur %7.1f end %8.1f" % (r[0].split('(')[0][-44:], qs.index(r[3]) + 1, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3))
PY
cat gpurun_out/r04_z/dhfr_timeline.txt
