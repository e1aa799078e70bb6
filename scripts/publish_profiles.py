"""copy an evidence set gpurun_out/<tag>/ into profiles/ under the naming of the earlier rounds: <tag>_<name>[_<ms>ms].json.log for bench lines
(the ms/step in the file name), <tag>_<name>.txt / .json for the summaries.  usage: python scripts/publish_profiles.py r06_z"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
SKIP = {"bench_default.err", "bench_under_rocprof.log", "pmc_fetch.log", "pmc_write.log", "pmc_mfma.log", "pmc_mix1.log", "pmc_mix2.log"}
for f in sorted(os.listdir(src)):
    if f in SKIP:
        continue
    p = os.path.join(src, f)
    if f.startswith("bench_") and (f.endswith(".json") or f == "bench_default.log"):
        try:
            line = [l for l in open(p).read().splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
        except Exception:      # noqa: BLE001
            print("skipped (no JSON line):", f)
            continue
        ms = d["ms_per_step"]
        extra = ""
        ip = d.get("input_pipeline")
        if f.startswith("bench_with_input_pipeline") and ip:
            extra = "_resident_%.3fms_fed" % ip["ms_per_step"]
        name = "%s_%s_%.3fms%s.json.log" % (tag, f.rsplit(".", 1)[0], ms, extra)
        open(os.path.join(dst, name), "w").write(line + "\n")
    else:
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, f)))
print("published", tag)
