#!/usr/bin/env python
"""per-kernel statistics out of a rocprofv3 rocpd SQLite database (the default output of `rocprofv3 --kernel-trace` in ROCm 7.2):
   python tools/rocpd_stats.py results.db [n_last_scenes_divisor] -> name, calls, total ms, avg us, % (sorted by total)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute("select s.kernel_name, count(*), sum(d.end - d.start), min(d.start), max(d.end) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name" % (kd, ks)).fetchall()
tot = sum(r[2] for r in rows)
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n.replace('pst::', '')[:110]


print('%-110s %8s %10s %9s %6s' % ('kernel', 'calls', 'total ms', 'avg us', '%'))
for name, n, t, _, _ in sorted(rows, key=lambda r: -r[2])[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print('%-110s %8.1f %10.2f %9.1f %6.1f' % (short(name), n / div, t / 1e6 / div, t / 1e3 / n, 100.0 * t / tot))
print('total kernel time %.1f ms (/%g = %.1f ms)' % (tot / 1e6, div, tot / 1e6 / div))
