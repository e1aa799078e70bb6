"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a --stats style table.
usage: python scripts/rocpd_summary.py gpurun_out/prof/x_results.db [steps] > profiles/NAME.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
scol = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
namecol = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[1])
q = "select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (namecol, kd, ks, namecol)
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows)
# one stem-conv forward launch per training step (graph replays, eager warm-up, capture, per-kernel timing passes alike)
steps = float(sum(r[1] for r in rows if "stem_conv_fwd" in r[0]) or (float(sys.argv[2]) if len(sys.argv) > 2 else 1.0))
print("# rocprofv3 --kernel-trace summary of %s (durations in us; per-step = total / %g training steps of any kind: replays, eager warm-up, capture, timing passes)" % (sys.argv[1], steps))
print("%-110s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for name, n, t, mn, mx in rows:
    name = re.sub(r"\s+", " ", name)
    if len(name) > 108:
        name = name[:105] + "..."
    print("%-110s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (name, n, t / 1e3, t / 1e3 / n, mn / 1e3, mx / 1e3, 100.0 * t / tot))
print("# total kernel time %.3f ms over %d dispatches; %.3f ms per step" % (tot / 1e6, sum(r[1] for r in rows), tot / 1e6 / steps))
