"""Kernel timeline (start, duration, queue) of two MD steps from a rocprofv3 rocpd database, anchored on the integrator chain
launches (round 3: the mesh pipeline no longer has a binning launch to anchor on).  usage: timeline2.py <dir> [first_chain_index]"""
import glob
import sqlite3
import sys
db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if r[0].startswith("integrate_chain")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
a, b = idx[k], idx[k + 2]
t0 = rows[a][1]
print("step wall us", (rows[idx[k + 200]][1] - rows[idx[k]][1]) / 200 / 1e3)
for r in rows[a:b + 1]:
    print("  %-45s q%-3d start %8.1f dur %7.1f end %8.1f" % (r[0][:45], r[3] % 1000, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3))
