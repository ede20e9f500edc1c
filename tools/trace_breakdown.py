"""steady-state per-callback kernel time from a rocprofv3 --kernel-trace database (last 10 callbacks)"""
import sqlite3, collections, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x * grid_y * grid_z from kernels order by start").fetchall()
key = lambda n: n.split('(')[0].replace('void aslam::', '')[:48]
gmax = max(g for n, s, e, g in rows if 'frontend' in n)   # the full-batch launches (bench.py also runs a batch-1 leg at the end)
fe = [(s, e) for n, s, e, g in rows if 'frontend' in n and g == gmax]
s0, s1 = fe[-11][0], fe[-1][0]
agg = collections.Counter(); cnt = collections.Counter()
for n, s, e, g in rows:
    if s0 <= s < s1:
        agg[key(n)] += (e - s) / 10 / 1e3; cnt[key(n)] += 1
print('callback period %.1f us, busy %.1f us' % ((s1 - s0) / 10 / 1e3, sum(agg.values())))
for n, v in agg.most_common():
    print('   %-50s %8.1f us/callback  (%d launches, %.1f us each)' % (n, v, cnt[n] // 10, v / (cnt[n] / 10)))
