"""Kernel concurrency report of a rocprofv3 rocpd database: dispatches per (queue, stream), summed kernel time,
wall time covered by at least one kernel, and time during which two or more kernels overlap."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
kd = [r[0] for r in cur.execute("select name from sqlite_master where type='table'") if "kernel_dispatch" in r[0]][0]
rows = list(cur.execute("select start,end,queue_id,stream_id from %s order by start" % kd))
print("dispatches", len(rows), dict(collections.Counter((r[2], r[3]) for r in rows)))
tot = busy = ov = 0
last = rows[0][0]
for s, e, _, _ in rows:
    tot += e - s
    if s < last:
        ov += min(e, last) - s
    if e > last:
        busy += e - max(s, last)
        last = e
print("sum of kernel durations %.2f ms, busy %.2f ms, overlapped %.2f ms" % (tot / 1e6, busy / 1e6, ov / 1e6))
