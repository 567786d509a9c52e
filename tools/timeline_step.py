"""Kernel timeline of a few consecutive MD steps from a rocprofv3 --kernel-trace rocpd database (anchor: integrate_chain_kernel).
usage: python tools/timeline_step.py <dir> [first anchor index from the end, default 60] [steps, default 2]"""
import glob
import sqlite3
import sys
db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'integrate_chain_kernel' in r[0]]
k = len(idx) - (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
a, b = idx[k], idx[k + n]
t0 = rows[a][1]
qs = sorted(set(r[3] for r in rows[a:b + 1]))
print("wall us per step", (rows[b][1] - rows[a][1]) / 1e3 / n)
for r in rows[a:b + 1]:
    print("  %-44s q%d start %8.1f dur %7.1f end %8.1f" % (r[0].split('(')[0][-44:], qs.index(r[3]) + 1, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3))
# steady-state average over the last 300 steps
a, b = idx[-320], idx[-20]
print("average over 300 steps: %.1f us per step" % ((rows[b][1] - rows[a][1]) / 1e3 / 300))
