"""List, for the last replayed step of a rocpd database, every launch of kernels matching PATTERN with its time offset, duration and the
names of the two launches before / one after: python scripts/rocpd_context.py DB PATTERN [max]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % ks)}
rows = list(cur.execute("select start,end,queue_id,kernel_id,grid_size_x,workgroup_size_x from %s order by start" % kd))
idx = [i for i, r in enumerate(rows) if "stem_conv_fwd" in names[r[3]]]
step = rows[idx[-1]:]
t0 = step[0][0]
short = lambda k: names[k].split("(")[0].replace("void ", "")[:44]
hits = 0
for i, r in enumerate(step):
    if sys.argv[2] in names[r[3]]:
        hits += 1
        if hits <= (int(sys.argv[3]) if len(sys.argv) > 3 else 400):
            prev2 = short(step[i - 2][3]) if i > 1 else "-"
            prev1 = short(step[i - 1][3]) if i > 0 else "-"
            nxt = short(step[i + 1][3]) if i + 1 < len(step) else "-"
            print("%8.3f ms %6.2f us grid %7d | %s | %s | -> %s" % ((r[0] - t0) / 1e6, (r[1] - r[0]) / 1e3, r[4], prev2, prev1, nxt))
print("total", hits)
