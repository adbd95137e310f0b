#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (SQLite) kernel trace: per-kernel calls / total / avg / min / max / %.
Usage: tools/rocpd_stats.py results.db [> profiles/<name>.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'fsmg::\(anonymous namespace\)::', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'\(fsmg::\w+Args(, unsigned long long\*)?\)', '', name)
    return name if len(name) <= 110 else name[:107] + '...'


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    # grouped by kernel AND grid so the shapes of one templated kernel (e.g. the dW / dKh / dKx GEMMs) stay apart
    rows = db.execute("select %s || '  [grid ' || grid_x || 'x' || grid_y || ', wg ' || workgroup_x || ']', count(*), "
                      "sum(end - start), avg(end - start), min(end - start), max(end - start) "
                      "from kernels group by %s, grid_x, grid_y order by 3 desc" % (name_col, name_col)).fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    print('%-110s %8s %12s %11s %11s %11s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for n, c, s, a, mn, mx in rows:
        print('%-110s %8d %12.1f %11.2f %11.2f %11.2f %6.2f' % (short(n), c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / total))
    print('%-110s %8d %12.1f' % ('TOTAL', sum(r[1] for r in rows), total / 1e3))


if __name__ == '__main__':
    main(sys.argv[1])
