"""MFMA utilisation per kernel family from a rocprofv3 --pmc database holding SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA, SQ_BUSY_CU_CYCLES
(or GRBM_GUI_ACTIVE): util = MFMA busy cycles / (SIMDs x elapsed shader cycles).  The elapsed cycles of a dispatch are taken from its
duration x the 2.4 GHz peak clock the dense-MFMA peak figure assumes (GRBM_GUI_ACTIVE / duration is printed for reference).
usage: rocpd_mfma_util.py pmc.db [out.json]"""
import collections
import json
import re
import sqlite3
import sys

SIMDS = 256 * 4
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, ip = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
scol = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
namecol = "display_name" if "display_name" in scol else "kernel_name"
q = ("select s.%s, p.name, d.id, e.value, d.end - d.start from %s e join %s p on e.pmc_id = p.id "
     "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id" % (namecol, pe, ip, kd, ks))
per = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(dict)
for name, ctr, did, v, dur in cur.execute(q):
    name = re.sub(r"\s+", " ", name).replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    per[name][ctr] += v
    disp[name][did] = dur
rows = []
for name, c in per.items():
    n = len(disp[name])
    ns = sum(disp[name].values())
    busy, insts, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_INSTS_MFMA", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    if insts <= 0:
        continue
    ghz = gui / ns if gui > 0 and ns > 0 else 0.0          # reported only: per-dispatch GRBM_GUI_ACTIVE includes profiling overhead
    cycles = ns * 2.4                                       # utilisation is quoted against the 2.4 GHz peak clock (what the dense-MFMA peak assumes)
    rows.append((ns, name, n, busy, insts, ghz, busy / (SIMDS * cycles) if cycles else 0.0))
rows.sort(reverse=True)
print("%-58s %6s %10s %14s %12s %9s %7s %9s" % ("kernel", "calls", "total_us", "mfma_busy_cyc", "mfma_insts", "cyc/inst", "gui/ns", "mfma_util"))
out = {}
for ns, name, n, busy, insts, ghz, util in rows:
    print("%-58s %6d %10.1f %14.0f %12.0f %9.2f %7.2f %8.2f%%" % (name[:58], n, ns / 1e3, busy, insts, busy / insts, ghz, 100 * util))
    out[name] = {"calls": n, "total_us": ns / 1e3, "mfma_busy_cycles": busy, "mfma_insts": insts, "mfma_util": util, "ghz": ghz}
tot_ns = sum(sum(d.values()) for d in disp.values())
tot_busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for c in per.values())
print("all kernels with MFMA: busy %.3e cycles; whole trace %.1f us of kernel time -> %.2f %% of SIMD-cycles at 2.4 GHz" % (
    tot_busy, tot_ns / 1e3, 100 * tot_busy / (SIMDS * tot_ns * 2.4)))
if len(sys.argv) > 2:
    json.dump({"kernels": out}, open(sys.argv[2], "w"), indent=1)
