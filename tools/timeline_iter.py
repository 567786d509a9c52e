"""Kernel timeline of ONE sampler iteration from a rocprofv3 rocpd database: everything between two launches of an anchor
kernel (default: the swap-all kernel).  usage: timeline_iter.py <dir> [anchor substring] [which anchor]"""
import glob
import sqlite3
import sys
db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
anchor = sys.argv[2] if len(sys.argv) > 2 else 'mix_swap_all'
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) - 2
a, b = idx[k], idx[k + 1]
t0 = rows[a][1]
print("iteration wall us", (rows[b][1] - rows[a][1]) / 1e3, "kernels", b - a)
prev_end = rows[a][1]
for r in rows[a:b + 1]:
    print("  %-60s q%-3d start %9.1f dur %8.1f gap_before %7.1f" % (r[0][:60], r[3] % 1000, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3))
    prev_end = max(prev_end, r[2])
