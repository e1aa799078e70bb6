"""Generate tests/golden/input_pipeline.npz -- run in the BUILD CONTAINER only (needs /root/reference and Pillow).

What it pins (see oracle/input_pipeline_ref.py for the leg-by-leg statement):
  * ``resize_*``: ``PIL.Image.resize`` outputs (the reference's ``loadvideo`` call, datasets/ava_frame.py:146-150) for a few
    (in, out) sizes -> pins ``oracle.input_pipeline_ref.pil_resize`` and the HIP ``tuber_frames_resize`` bit-exactly.
  * ``train_*`` / ``val_*``: the reference's OWN ``make_transforms('train'|'val')`` pipelines (datasets/ava_frame.py:158-176 ->
    datasets/video_transforms.py) run on PIL frames under ``random.seed(s)``: final clip tensor (3,T,h,w), boxes / raw_boxes / labels /
    size / area.  This pins the random draw order, the flip/crop composition and all box bookkeeping.
    Harness stand-ins, because torchvision and cv2 are not in this image: ``torchvision.transforms.functional`` crop / hflip (pure PIL
    indexing), to_tensor / normalize (documented fp32 arithmetic), and ``cv2.cvtColor`` = this repo's restatement of OpenCV's 8-bit
    HSV conversions -- so the ColorJitter ARITHMETIC is NOT pinned by these vectors (stated as "parity unpinned"), only where it sits in
    the pipeline and the draws it makes.
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import input_pipeline_ref as R            # noqa: E402
from oracle import ref_import                          # noqa: E402


def install_image_shims():
    from PIL import Image
    ref_import.install_shims()
    mod = ref_import._mod

    def to_tensor(pic):
        a = np.asarray(pic)
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div(255)

    def normalize(t, mean, std):
        m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
        s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
        return (t - m) / s

    F = mod("torchvision.transforms.functional",
            crop=lambda img, i, j, h, w: img.crop((j, i, j + w, i + h)),
            hflip=lambda img: img.transpose(Image.FLIP_LEFT_RIGHT),
            resize=lambda img, size: img.resize(size[::-1]),
            pad=None, to_tensor=to_tensor, normalize=normalize)
    T = mod("torchvision.transforms", functional=F, RandomCrop=object, RandomErasing=object)
    sys.modules["torchvision"].transforms = T
    cv2 = sys.modules["cv2"]
    cv2.COLOR_RGB2HSV, cv2.COLOR_HSV2RGB = "rgb2hsv", "hsv2rgb"
    cv2.cvtColor = lambda a, code: R.rgb2hsv_u8(np.asarray(a)) if code == "rgb2hsv" else R.hsv2rgb_u8(np.asarray(a))


def smooth_frames(rng, T, H, W):
    """frames with structure at several scales (pure noise would hide resampling-phase mistakes less well than edges do)."""
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.empty((T, H, W, 3), np.uint8)
    for t in range(T):
        img = np.zeros((H, W, 3))
        for c in range(3):
            fx, fy, ph = rng.uniform(0.02, 0.3), rng.uniform(0.02, 0.3), rng.uniform(0, 6.28)
            img[..., c] = 127 + 90 * np.sin(fx * xx + fy * yy + ph + 0.1 * t) + rng.normal(0, 12, (H, W))
        img[H // 4:H // 2, W // 3:W // 2] = rng.integers(0, 256, 3)
        out[t] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def make_target(rng, nh, nw, nbox):
    """same fields / dtypes as ``load_annotation`` (datasets/ava_frame.py:97-129)."""
    x1 = rng.integers(0, nw // 2, nbox); y1 = rng.integers(0, nh // 2, nbox)
    x2 = x1 + rng.integers(4, nw // 2, nbox); y2 = y1 + rng.integers(4, nh // 2, nbox)
    boxes = torch.as_tensor(np.stack([np.full(nbox, 4), x1, y1, x2, y2], 1), dtype=torch.float32)
    boxes[:, 1::3].clamp_(min=0, max=int(nw))
    boxes[:, 2::3].clamp_(min=0, max=nh)
    raw = torch.nn.functional.pad(boxes, (1, 0, 0, 0), value=7)
    labels = torch.as_tensor(rng.integers(0, 2, (nbox, 80)), dtype=torch.float32)
    return {"image_id": ["vid_0902", 4], "boxes": boxes, "raw_boxes": raw, "labels": labels,
            "orig_size": torch.as_tensor([int(nh), int(nw)]), "size": torch.as_tensor([int(nh), int(nw)])}


def main():
    from PIL import Image
    install_image_shims()
    rng = np.random.default_rng(20260928)
    g = {}
    # ---- Pillow resize vectors
    cases = [(60, 80, 48, 64), (45, 80, 72, 128), (64, 48, 64, 30), (50, 70, 77, 70), (90, 160, 36, 64)]
    g["resize_cases"] = np.array(cases)
    for k, (H, W, oh, ow) in enumerate(cases):
        a = smooth_frames(rng, 2, H, W)
        g["resize_in_%d" % k] = a
        g["resize_out_%d" % k] = np.stack([np.asarray(Image.fromarray(f).resize((ow, oh))) for f in a])
    # ---- the reference's transform pipelines
    with ref_import.reference_on_path():
        import datasets.ava_frame as ava
        cfg = ref_import._to_node({"CONFIG": {"DATA": {"IMG_SIZE": 40}}})
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            tf = {"train": ava.make_transforms("train", cfg), "val": ava.make_transforms("val", cfg)}
        samples = [("train", 11, 48, 64), ("train", 12, 48, 64), ("train", 17, 66, 45), ("train", 20, 40, 71),
                   ("val", 15, 48, 64), ("val", 16, 66, 45)]
        g["sample_meta"] = np.array([(0 if m == "train" else 1, s, nh, nw) for m, s, nh, nw in samples])
        for k, (mode, seed, nh, nw) in enumerate(samples):
            H0, W0 = nh + 9, nw + 12                      # decoded size differs from (nh, nw): loadvideo resizes every frame
            frames = smooth_frames(rng, 4, H0, W0)
            target = make_target(rng, nh, nw, 5)
            g["s%d_frames" % k] = frames
            for f in ("boxes", "raw_boxes", "labels"):
                g["s%d_in_%s" % (k, f)] = target[f].numpy().copy()
            imgs = [Image.fromarray(f).resize((nw, nh)) for f in frames]      # ava_frame.py:146-150
            random.seed(seed)
            imgs, tgt = tf[mode](imgs, target)
            clip = torch.stack(imgs, dim=0).permute(1, 0, 2, 3)               # ava_frame.py:71-72
            g["s%d_clip" % k] = clip.numpy()
            for f in ("boxes", "raw_boxes", "labels", "size", "area"):
                g["s%d_out_%s" % (k, f)] = tgt[f].numpy()
            print(mode, seed, "clip", tuple(clip.shape), "boxes kept", len(tgt["boxes"]))
    out = os.path.join(ROOT, "tests", "golden", "input_pipeline.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
