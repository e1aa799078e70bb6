import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import input_pipeline_ref as R
from tubelet_transformer_amd import input_pipeline as P
dev = torch.device("cuda:0")
lv = np.arange(0, 256, 5, dtype=np.uint8)
grid = np.stack(np.meshgrid(lv, lv, lv, indexing="ij"), -1).reshape(1, 52, 52 * 52, 3)
lut = R.normalize_lut()
for jit in [None, (0, 0, 0), (10, 26, 26), (-9, 11, 0)]:
    c = P.FrameClip(grid.copy()); c.jitter = jit
    got = P.ClipBatch([c]).to(dev).tensors.cpu().numpy()[0]
    want = R.prepare_clip(grid, jitter=jit)
    bad = np.argwhere(got != want)
    print("jitter", jit, "mismatches", len(bad), "of", got.size)
    for b in bad[:6]:
        c_, t, y, x = b
        px = grid[t, y, x]
        inv = {float(v): i for i, v in enumerate(lut[c_])}
        print("   rgb", px, "chan", c_, "want u8", inv.get(float(want[tuple(b)])), "got u8", inv.get(float(got[tuple(b)])), "hsv", R.rgb2hsv_u8(px[None, None])[0, 0])
