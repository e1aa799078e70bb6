"""Average of every collected PMC counter per kernel (rocprofv3 --pmc ... rocpd database).  usage: rocpd_pmc_generic.py db [filter]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, ip = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
scol = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
namecol = "display_name" if "display_name" in scol else "kernel_name"
q = ("select s.%s, d.grid_size_x, p.name, count(*), avg(e.value), avg(d.end - d.start) from %s e join %s p on e.pmc_id = p.id "
     "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.%s, d.grid_size_x, p.name" % (namecol, pe, ip, kd, ks, namecol))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for name, gx, ctr, n, v, dur in cur.execute(q):
    name = re.sub(r"\s+", " ", name)
    if flt in name:
        print("%-70s grid %-9d %-22s n=%-4d avg %14.3f   (%.1f us)" % (name[:70], gx, ctr, n, v, dur / 1e3))
