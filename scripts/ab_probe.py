"""print the per-tensor gradient differences of every TUBER_AB switch against the default path (tolerance calibration of the A/B test)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_training_gpu import _ab_run, _ab_names
dev = torch.device("cuda:0")
l0, g0, b0, _ = _ab_run(dev, ())
for name in ["(default again)"] + _ab_names():
    l1, g1, b1, _ = _ab_run(dev, () if name.startswith("(") else (name,))
    rels = sorted(((float((g1[n] - g0[n]).norm()) / (float(g0[n].norm()) + 1e-30), n) for n in g0), reverse=True)
    v = [r for r, _ in rels]
    print("%-28s loss %.5f vs %.5f  median %.2e  p90 %.2e  p99 %.2e  max %.2e  >0.1: %d of %d   worst: %s" % (
        name, l1, l0, v[len(v) // 2], v[len(v) // 10], v[len(v) // 100], v[0], sum(x > 0.1 for x in v), len(v),
        ", ".join("%s %.2f" % (n.replace("backbone.body.", ""), r) for r, n in rels[:4])), flush=True)
