"""Summarises a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (markdown/CSV)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(grid_y), max(grid_z), max(workgroup_x) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ['| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds | grid | wg |',
             '|---|---|---|---|---|---|---|---|---|---|---|---|']
    for r in rows:
        name = r[0].split('(')[0][:70]
        lines.append('| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f | %s | %s | %s | %sx%sx%s | %s |' % (
            name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
    text = '\n'.join(lines) + '\n\ntotal kernel time %.3f ms\n' % (total / 1e6)
    if out:
        open(out, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
