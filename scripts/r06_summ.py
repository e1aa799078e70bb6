"""one line per bench JSON log of a directory: ms/step, exposed ms, issue points, graph part durations"""
import glob
import json
import os
import sys

for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:      # noqa: BLE001
        print("%-40s unreadable (%s)" % (os.path.basename(f)[:-5], e))
        continue
    c = d.get("comm") or {}
    pts = [(p["issued_at_ms"], p["MB"]) for p in (c.get("issue_points") or [])]
    print("%-40s %8.3f ms  exposed %s  issue points (ms after step begin, MB) %s  graph parts ms %s" % (
        os.path.basename(f)[:-5], d["ms_per_step"], c.get("exposed_ms") and round(c["exposed_ms"], 3), pts, c.get("graph_part_ms")))
