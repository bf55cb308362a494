"""Summarise a rocprofv3 --pmc run (rocpd sqlite): per kernel, summed counters of the dispatches issued AFTER the last
Adam launch (i.e. the per-layer replays of tools/layer_times.py, not the warm-up training steps)."""
import glob
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
cur = db.cursor()
t0 = 0 if len(sys.argv) > 3 else (list(cur.execute("select max(end) from counters_collection where kernel_name like '%adam%'"))[0][0] or 0)
rows = list(cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(end-start)/1e6 from counters_collection "
                        "where start > ? group by kernel_name, counter_name", (t0,)))
d = defaultdict(dict)
t = {}
for k, c, v, n, ms in rows:
    d[k][c] = v
    t[k] = (n, ms)
for k, v in sorted(d.items(), key=lambda kv: -t[kv[0]][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(k[:110], 'calls', t[k][0], 'ms %.2f' % t[k][1])
    print('    ', {c: '%.4g' % x for c, x in v.items()})
