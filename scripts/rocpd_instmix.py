"""Per kernel family: wave-level instruction counts per launch (VALU, MFMA, LDS, SALU, VMEM), VALU per MFMA, and the LDS bank-conflict
fraction (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE cycles), from two rocprofv3 --pmc databases.
usage: rocpd_instmix.py pass1.db pass2.db"""
import collections
import re
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    kd, ks, pe, ip = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    scol = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
    namecol = "display_name" if "display_name" in scol else "kernel_name"
    q = ("select s.%s, p.name, count(*), sum(e.value), sum(d.end - d.start) from %s e join %s p on e.pmc_id = p.id "
         "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.%s, p.name" % (namecol, pe, ip, kd, ks, namecol))
    out = collections.defaultdict(dict)
    for name, ctr, n, v, dur in cur.execute(q):
        name = re.sub(r"\s+", " ", name).replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        out[name][ctr] = (n, v, dur)
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k, d in a.items():
    if "SQ_INSTS_VALU" not in d:
        continue
    n, valu, dur = d["SQ_INSTS_VALU"]
    mfma = d.get("SQ_INSTS_MFMA", (0, 0, 0))[1]
    lds = d.get("SQ_INSTS_LDS", (0, 0, 0))[1]
    salu = d.get("SQ_INSTS_SALU", (0, 0, 0))[1]
    e = b.get(k, {})
    conf = e.get("SQ_LDS_BANK_CONFLICT", (0, 0, 0))[1]
    act = e.get("SQ_LDS_IDX_ACTIVE", (0, 0, 0))[1]
    rd = e.get("SQ_INSTS_VMEM_RD", (0, 0, 0))[1]
    wr = e.get("SQ_INSTS_VMEM_WR", (0, 0, 0))[1]
    rows.append((dur, k, n, valu / n, mfma / n, lds / n, salu / n, rd / max(e.get("SQ_INSTS_VMEM_RD", (1,))[0], 1), wr / max(e.get("SQ_INSTS_VMEM_WR", (1,))[0], 1),
                 valu / mfma if mfma else float("nan"), conf / act if act else float("nan")))
print("# wave-instructions per launch, averaged over the launches of the run (one warm-up + capture + one replay of the BASELINE step)")
print("%-58s %6s %10s %9s %9s %9s %9s %9s %9s %8s" % ("kernel", "calls", "VALU", "MFMA", "LDS", "SALU", "VMEM_RD", "VMEM_WR", "VALU/MFMA", "LDSconf"))
for dur, k, n, valu, mfma, lds, salu, rd, wr, ratio, cf in sorted(rows, reverse=True)[:60]:
    print("%-58s %6d %10.0f %9.0f %9.0f %9.0f %9.0f %9.0f %9.1f %8.3f" % (k[:58], n, valu, mfma, lds, salu, rd, wr, ratio, cf))
