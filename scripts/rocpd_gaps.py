"""Gaps between consecutive kernels (no kernel running on any queue) above a threshold, for the last N replayed steps of a rocpd
database: python scripts/rocpd_gaps.py DB [min_us=10] [steps=4]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % ks)}
rows = list(cur.execute("select start,end,queue_id,kernel_id from %s order by start" % kd))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 4
idx = [i for i, r in enumerate(rows) if "stem_conv_fwd" in names[r[3]]]
short = lambda k: names[k].split("(")[0].replace("void ", "")[:40]
for si in range(len(idx) - nst, len(idx)):
    a = idx[si]
    b = idx[si + 1] if si + 1 < len(idx) else len(rows)
    step = rows[a:b]
    t0 = step[0][0]
    last = step[0][1]
    tot = 0.0
    out = []
    for j in range(1, len(step)):
        s, e, q, k = step[j]
        if s > last:
            g = (s - last) / 1e3
            tot += g
            if g >= thr:
                out.append("%.2fms:+%.0fus(%s->%s)" % ((s - t0) / 1e6, g, short(step[j - 1][3])[:18], short(k)[:18]))
        last = max(last, e)
    print("step %d: %d kernels, span %.3f ms, idle %.3f ms; gaps>=%.0fus: %s" % (si, len(step), (step[-1][1] - t0) / 1e6, tot / 1e3, thr, " ".join(out[:12])))
