"""Volume of the deferred weight-gradient partials of one BASELINE training step (what tuber_multi_reduce reads)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tubelet_transformer_amd import synth
from tubelet_transformer_amd.config import load_cfg
from tubelet_transformer_amd.tuber import build_model
from tubelet_transformer_amd.training import train_step, build_optimizer
dev = torch.device("cuda:0")
cfg = load_cfg(os.path.join(ROOT, "configuration", "TubeR_CSN152_AVA21.yaml"))
model, crit, _ = build_model(cfg)
synth.load_name_hashed(model)
model.to(dev).train(); crit.to(dev).train()
opt = build_optimizer(model, cfg)
clips = synth.synthetic_clips(2, 32, 256, 340, seed=1, device=dev)
targets = synth.synthetic_targets(2, "ava", 80, seed=2, device=dev, hw=(256, 340))
train_step(model, crit, opt, clips, targets, 0.1, cfg=cfg)
torch.cuda.synchronize()
store, _ = model.engine()
E = store.defer._ENTRY
for key, (tab, blk, nb) in store.defer.cache.items():
    t = np.frombuffer(tab.cpu().numpy().tobytes(), dtype=E)
    by = (t["S"].astype(np.int64) * t["n"] * 4).sum()
    print("table: %d entries, %d blocks, partial bytes %.1f MB, output bytes %.1f MB; by mode:" % (len(t), nb, by / 1e6, (t["n"] * 4).sum() / 1e6),
          {int(m): "%.1f MB" % ((t["S"].astype(np.int64) * t["n"] * 4)[t["mode"] == m].sum() / 1e6) for m in np.unique(t["mode"])},
          "S histogram", dict(zip(*np.unique(t["S"][t["mode"] == 2], return_counts=True))))
