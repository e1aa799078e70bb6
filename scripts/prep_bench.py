"""Time the clip input pre-pass (tuber_frames_resize + tuber_clip_prepare) on the BASELINE batch: 2 clips of 32 frames decoded at
360x480, resized to 288x384 (IMG_RESHAPE_SIZE), flipped, cropped to 256x340 (IMG_SIZE), colour-jittered, normalised, collated.
Prints per-kernel time (HIP events around 20 launches each, inputs resident in HBM) and the algorithmic GB/s of each."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubelet_transformer_amd import input_pipeline as P, lib   # noqa: E402

dev = torch.device("cuda:0")
T, H0, W0, H, W, h, w = 32, 360, 480, 288, 384, 256, 340
N = 2
rng = np.random.default_rng(0)
src = [torch.from_numpy(rng.integers(0, 256, (T, H0, W0, 3), dtype=np.uint8)).to(dev) for _ in range(N)]
(bh, kh, bv, kv), ksh, ksv, y0, rows = P._device_coeffs(dev, H0, W0, H, W)
per = (T * H * W * 3 + 255) // 256 * 256
staging = torch.empty(N * per, dtype=torch.uint8, device=dev)
tmp = torch.empty(T * rows * W * 3, dtype=torch.uint8, device=dev)
desc = np.zeros(N, P._DESC)
for i in range(N):
    desc[i] = (i * per, H, W, 16, 22, h, w, i % 2, 1, 7, -13, 20, 0)
ddesc = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
lut, hsv = P._device_tables(dev, (P.MEAN, P.STD))
out = torch.empty(N, 3, T, h, w, dtype=torch.float32, device=dev)
mask = torch.empty(N, h, w, dtype=torch.bool, device=dev)


def resize():
    for i in range(N):
        lib.call("tuber_frames_resize", src[i], tmp, staging[i * per:], T, H0, W0, H, W, bh, kh, ksh, bv, kv, ksv, y0, rows)


def prepare():
    lib.call("tuber_clip_prepare", staging, ddesc, lut, hsv, out, mask, N, T, h, w)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


t_r = timeit(resize)
t_p = timeit(prepare)
b_r = N * T * (rows * W0 * 3 + rows * W * 3 * 2 + H * W * 3)            # h pass reads rows of the source, writes tmp; v pass reads tmp, writes dst
b_p = N * T * h * w * (3 + 12) + N * h * w
print("resize 2x32x%dx%d -> %dx%d: %.1f us (two launches per clip), %.2f GB algorithmic -> %.0f GB/s" % (H0, W0, H, W, t_r, b_r / 1e9, b_r / t_r / 1e3))
print("prepare (flip+crop+jitter+normalise+collate) -> 2x3x32x%dx%d fp32: %.1f us, %.3f GB algorithmic -> %.0f GB/s" % (h, w, t_p, b_p / 1e9, b_p / t_p / 1e3))
desc["jitter"] = 0
ddesc.copy_(torch.from_numpy(desc.view(np.uint8).copy()).to(dev))
print("prepare without jitter: %.1f us" % timeit(prepare))
host = [s.cpu().numpy() for s in src]
import time
clips = []
for i in range(N):
    c = P.FrameClip(host[i]).resize((W, H)); t = {"labels": torch.zeros(0)}
    P.hflip(c, t); P.crop(c, t, (16, 22, h, w)); c.jitter = (7, -13, 20); clips.append(c)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    nt = P.ClipBatch(clips).to(dev)
torch.cuda.synchronize()
print("ClipBatch.to(device) from pageable host uint8 frames (33 MB H2D + pre-pass): %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
