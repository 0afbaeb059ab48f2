"""Summarise a rocprofv3 rocpd database (the `*_results.db` a `--kernel-trace` run leaves): per-kernel calls / avg / total.

    python tools/prof_db.py gpurun_out/prof/x_results.db [substring ...]"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute('select name, start, end, grid_x, workgroup_x from kernels order by start'))
    filt = sys.argv[2:]
    agg = collections.OrderedDict()
    for name, s, e, gx, wx in rows:
        nm = re.sub(r'\(.*', '', name).replace('void ', '').replace('pk::', '')
        if filt and not any(f in nm for f in filt):
            continue
        a = agg.setdefault(nm, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    tot = sum(a[1] for a in agg.values())
    print(f'total kernel time {tot / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} launches')
    for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f'{nm[:84]:84s} {a[0]:6d} avg {a[1] / a[0] / 1e3:8.1f} min {a[2] / 1e3:8.1f} max {a[3] / 1e3:8.1f} us {100.0 * a[1] / tot:5.1f}%')


if __name__ == '__main__':
    main()
