"""All kernels of every queue inside a time window of a rocprofv3 --kernel-trace rocpd database (several handles = several pairs of
queues).  usage: python tools/timeline_window.py <dir> [window us = 700] [anchor = the chain launch this many from the end = 150]"""
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 700.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 150
idx = [i for i, r in enumerate(rows) if r[0].startswith("integrate_chain")]
a = idx[-back]; t0 = rows[a][1]
qs = sorted(set(r[3] for r in rows))
n_chain = len(idx)
span = (rows[idx[-20]][1] - rows[idx[-320]][1]) / 1e3 if n_chain > 340 else float('nan')
print("queues", len(qs), " chain launches", n_chain, " us per chain launch over 300 launches: %.1f" % (span / 300.0))
for r in rows[a:]:
    s = (r[1] - t0) / 1e3
    if s > win: break
    print("  %-40s q%-2d start %8.1f dur %7.1f end %8.1f" % (r[0].split('(')[0].replace('void ', '')[:40], qs.index(r[3]), s, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3))
