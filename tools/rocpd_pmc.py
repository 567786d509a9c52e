"""Summarises rocprofv3 --pmc runs (rocpd sqlite) into per-kernel average counter values per dispatch."""
import json
import sqlite3
import sys


def main(paths, out_md=None, out_json=None):
    table = {}
    for path in paths:
        db = sqlite3.connect(path)
        for name, counter, n, avg in db.execute(
                "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            table.setdefault(name.split('(')[0], {})[counter] = (n, avg)
    lines = ['| kernel | dispatches | FETCH_SIZE KB/dispatch | x2 (gfx950 wide-read correction) MB | WRITE_SIZE KB/dispatch | HBM-side MB/dispatch (2*fetch + write) |',
             '|---|---|---|---|---|---|']
    js = {}
    for k, v in sorted(table.items(), key=lambda kv: -(kv[1].get('FETCH_SIZE', (0, 0))[1] + kv[1].get('WRITE_SIZE', (0, 0))[1])):
        f = v.get('FETCH_SIZE', (0, 0.0)); w = v.get('WRITE_SIZE', (0, 0.0))
        total_mb = (2.0 * f[1] + w[1]) / 1024.0
        lines.append('| %s | %d | %.1f | %.2f | %.1f | %.2f |' % (k[:60], max(f[0], w[0]), f[1], 2 * f[1] / 1024.0, w[1], total_mb))
        js[k] = dict(fetch_kb=f[1], write_kb=w[1], hbm_mb_corrected=total_mb)
    text = '\n'.join(lines) + '\n'
    if out_md:
        open(out_md, 'w').write(text)
    if out_json:
        json.dump(js, open(out_json, 'w'), indent=1)
    print(text)


if __name__ == '__main__':
    args = sys.argv[1:]
    md = args[args.index('--md') + 1] if '--md' in args else None
    js = args[args.index('--json') + 1] if '--json' in args else None
    paths = [a for a in args if a.endswith('.db')]
    main(paths, md, js)
