"""Launch-by-launch listing (start offset, duration, gap to the previous kernel) of a window of the shortest training step (= a hipGraph
replay) in a rocprofv3 rocpd database.  usage: rocpd_sequence.py db start_ms end_ms"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % ks)}
rows = list(cur.execute("select start,end,kernel_id,grid_size_x,workgroup_size_x from %s order by start" % kd))
idx = [i for i, r in enumerate(rows) if "stem_conv_fwd" in names[r[2]]]
spans = [(rows[idx[j + 1] - 1][1] - rows[idx[j]][0], j) for j in range(len(idx) - 1)]
best = min(spans)[1]
step = rows[idx[best]:idx[best + 1]]
t0 = step[0][0]
a, b = float(sys.argv[2]) * 1e6 + t0, float(sys.argv[3]) * 1e6 + t0
prev = None
for s, e, k, gx, wx in step:
    if s < a:
        prev = e
        continue
    if s >= b:
        break
    n = names[k].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:64]
    print("%9.1f us  dur %7.2f  gap %6.2f  wgs %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, gx // max(wx, 1), n))
    prev = e
