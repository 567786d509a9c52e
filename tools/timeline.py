"""Prints the kernel timeline (start, duration, queue) of a few MD steps from a rocprofv3 rocpd database."""
import glob
import sqlite3
import sys
db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if r[0].startswith("pme_bin")]
a, b = idx[100], idx[102]
t0 = rows[a][1]
print("step wall us", (rows[idx[300]][1] - rows[idx[100]][1]) / 200 / 1e3)
for r in rows[a - 3:b]:
    print("  %-45s q%-3d start %8.1f dur %7.1f end %8.1f" % (r[0][:45], r[3] % 1000, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3))
