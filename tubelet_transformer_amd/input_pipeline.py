"""Clip input pipeline with the pixel work on the GPU.

The reference decodes, resizes, flips, crops, colour-jitters, normalises and pads every clip on CPU DataLoader workers and ships
67 MB of fp32 per step to the device (``datasets/ava_frame.py:37-74,133-176``, ``datasets/video_transforms.py``,
``utils/misc.py:279-282,367-425``).  Here the transforms keep the reference's names, call order, random draws and box
bookkeeping, but on the image side they only RECORD geometry on a :class:`FrameClip` (decoded uint8 frames + a plan); the pixels move
once, as uint8, and two HIP launches (``tuber_frames_resize``, ``tuber_clip_prepare`` -- ``csrc/clip_prep.hip``) produce the padded,
normalised fp32 batch and its mask directly in HBM when the training loop calls ``samples.to(device)`` -- the same call the
reference loop makes on its ``NestedTensor`` (``utils/video_action_recognition.py:121,280,513``).

There is no CPU pixel path in this module: ``ClipBatch.to`` needs the HIP library and a GPU.
"""
import random

import numpy as np
import torch

from . import lib
from .box_ops import box_xyxy_to_cxcywh
from .misc import NestedTensor

MEAN = (0.485, 0.456, 0.406)     # datasets/ava_frame.py:161
STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2      # Pillow Resample.c


# ----------------------------------------------------------------------------------------------------------------------
# host-side tables
# ----------------------------------------------------------------------------------------------------------------------
def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def resize_coeffs(in_size, out_size):
    """Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the bicubic filter, vectorised in float64 with the same operation
    order.  -> (bounds int32 [out,2] = (first tap, taps), kk int32 [out,ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    w = _bicubic(((x + xmin[:, None]) - center[:, None] + 0.5) * ss)
    w = np.where(x < xmax[:, None], w, 0.0)
    ww = np.cumsum(w, axis=1)[:, -1:]                      # left-to-right double sum like the C loop
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    q = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))
    kk = np.where(x < xmax[:, None], np.trunc(q), 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, kk


def normalize_lut(mean=MEAN, std=STD):
    """fp32 [3,256]: ToTensor (u8 / 255) then Normalize ((x - mean) / std), each step rounded to fp32 like torchvision."""
    u = np.arange(256, dtype=np.float32) / np.float32(255)
    return np.stack([(u - np.float32(m)) / np.float32(s) for m, s in zip(mean, std)]).astype(np.float32)


def hsv_tables():
    """OpenCV's 12-bit fixed-point division tables of the 8-bit RGB->HSV conversion: int32 [2,256] (sdiv, hdiv for hue range 180)."""
    i = np.arange(1, 256, dtype=np.float64)
    tab = np.zeros((2, 256), np.int32)
    tab[0, 1:] = np.rint((255 << 12) / (1.0 * i))
    tab[1, 1:] = np.rint((180 << 12) / (6.0 * i))
    return tab


_DESC = np.dtype([("src_off", "<i8"), ("H", "<i4"), ("W", "<i4"), ("y1", "<i4"), ("x1", "<i4"), ("h", "<i4"), ("w", "<i4"),
                  ("flip", "<i4"), ("jitter", "<i4"), ("hue", "<i4"), ("sat", "<i4"), ("val", "<i4"), ("pad", "<i4")])


# ----------------------------------------------------------------------------------------------------------------------
# the sample: decoded frames + deferred geometry
# ----------------------------------------------------------------------------------------------------------------------
class FrameClip:
    """Decoded frames of one sample, uint8 [T,H,W,3] (numpy, host), plus what the transforms decided to do with them.

    Stands where the reference has a list of ``PIL.Image`` (``imgs``): ``imgs[0].width`` / ``.height`` / ``.size`` work, so the
    transform code reads like the reference's.  Current view = source pixel (oy + y, ox + sx * x) for y < height, x < width.
    """

    def __init__(self, frames):
        frames = np.ascontiguousarray(frames)
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[3] != 3:
            raise ValueError("FrameClip wants uint8 frames [T,H,W,3], got %s %s" % (frames.dtype, frames.shape))
        self.frames = frames
        self.pinned = None               # page-locked torch view of the frames (ClipBatch.pin_memory: the DataLoader's pin thread)
        self.resize_hw = None
        self.oy, self.ox, self.sx = 0, 0, 1
        self.height, self.width = int(frames.shape[1]), int(frames.shape[2])
        self.jitter = None
        self.norm = None

    # list-of-PIL look-alike
    def __getitem__(self, i):
        return self

    def __len__(self):
        return int(self.frames.shape[0])

    @property
    def size(self):                      # PIL order
        return (self.width, self.height)

    @property
    def frame_hw(self):
        return self.resize_hw if self.resize_hw is not None else tuple(int(v) for v in self.frames.shape[1:3])

    def resize(self, size):
        """``PIL.Image.resize((w, h))`` of every frame (``ava_frame.py:148``); must precede flip / crop."""
        w, h = int(size[0]), int(size[1])
        if (self.oy, self.ox, self.sx) != (0, 0, 1) or (self.height, self.width) != self.frame_hw or self.resize_hw is not None:
            raise NotImplementedError("resize after another geometric transform (no published pipeline does this)")
        if (h, w) != self.frame_hw:
            self.resize_hw = (h, w)
        self.height, self.width = h, w
        return self

    def _hflip(self):
        self.ox += self.sx * (self.width - 1)
        self.sx = -self.sx

    def _crop(self, i, j, h, w):
        if i < 0 or j < 0 or i + h > self.height or j + w > self.width:
            raise ValueError("crop window (%d,%d,%d,%d) outside the %dx%d frame" % (i, j, h, w, self.height, self.width))
        self.oy += i
        self.ox += self.sx * j
        self.height, self.width = int(h), int(w)

    def descriptor(self, src_off):
        H, W = self.frame_hw
        flip = self.sx < 0
        x1 = (W - 1 - self.ox) if flip else self.ox
        jit = self.jitter or (0, 0, 0)
        return (src_off, H, W, self.oy, x1, self.height, self.width, int(flip), int(self.jitter is not None), jit[0], jit[1], jit[2], 0)


# ----------------------------------------------------------------------------------------------------------------------
# transforms: reference names and semantics (datasets/video_transforms.py); boxes are [:, 0] = frame index, [:, 1:] = xyxy
# ----------------------------------------------------------------------------------------------------------------------
def crop(images, target, region):
    """video_transforms.py:20-66 -- crop window (top, left, h, w); boxes shifted / clipped, boxes with area <= 30 dropped."""
    i, j, h, w = (int(v) for v in region)
    images._crop(i, j, h, w)
    target = target.copy()
    target["size"] = torch.tensor([h, w])
    fields = ["labels"]
    if "boxes" in target:
        boxes = target["boxes"][:, 1:]
        max_size = torch.as_tensor([w, h], dtype=torch.float32)
        cropped = boxes - torch.as_tensor([j, i, j, i])
        cropped = torch.min(cropped.reshape(-1, 2, 2), max_size).clamp(min=0)
        area = (cropped[:, 1, :] - cropped[:, 0, :]).prod(dim=1)
        target["boxes"][:, 1:] = cropped.reshape(-1, 4)
        target["raw_boxes"] = torch.cat((target["raw_boxes"][:, 0:1], target["boxes"]), 1)
        target["area"] = area
        fields += ["boxes", "raw_boxes"]
        keep = area > 30
        for f in fields:
            target[f] = target[f][keep]
    return images, target


def hflip(images, target):
    """video_transforms.py:69-85."""
    w, h = images.size
    images._hflip()
    target = target.copy()
    if "boxes" in target:
        boxes = target["boxes"][:, 1:]
        boxes = boxes[:, [2, 1, 0, 3]] * torch.as_tensor([-1, 1, -1, 1]) + torch.as_tensor([w, 0, w, 0])
        target["boxes"][:, 1:] = boxes
        target["raw_boxes"] = torch.cat((target["raw_boxes"][:, 0:1], target["boxes"]), 1)
    return images, target


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, imgs, target):
        if random.random() < self.p:
            return hflip(imgs, target)
        return imgs, target


class HorizontalFlip:
    def __call__(self, imgs, target):
        return hflip(imgs, target)


class RandomSizeCrop_Custom:
    """video_transforms.py:184-208: short side cropped to ``size`` (kept when already smaller), long side by the aspect ratio."""

    def __init__(self, size):
        self.size = size

    def __call__(self, imgs, target):
        W, H = imgs[0].width, imgs[0].height
        if W < H:
            w = W if W < self.size else self.size
            h = int(w * (H / W))
        else:
            h = H if H < self.size else self.size
            w = int(h * (W / H))
        x1 = random.randint(0, W - w)
        y1 = random.randint(0, H - h)
        return crop(imgs, target, (y1, x1, h, w))


class Resize_Custom:
    """video_transforms.py:210-227: despite the name a centre crop to (size * aspect) -- the reference's "fake crop"."""

    def __init__(self, size):
        self.size = size

    def __call__(self, imgs, target):
        W, H = imgs[0].width, imgs[0].height
        if W < H:
            w = self.size
            h = int(self.size * (H / W))
        else:
            h = self.size
            w = int(self.size * (W / H))
        top = int(round((H - h) / 2.0))
        left = int(round((W - w) / 2.0))
        return crop(imgs, target, (top, left, h, w))


class CenterCrop:
    def __init__(self, size):
        self.size = size

    def __call__(self, imgs, target):
        W, H = imgs[0].size
        ch, cw = self.size
        return crop(imgs, target, (int(round((H - ch) / 2.0)), int(round((W - cw) / 2.0)), ch, cw))


class ColorJitter:
    """video_transforms.py:333-369: one (hue, saturation, value) shift per clip in OpenCV's 8-bit HSV space; the three draws are made
    here in the reference's order, the conversion runs inside ``tuber_clip_prepare``."""

    def __init__(self, hue_shift=20.0, sat_shift=0.1, val_shift=0.1):
        self.hue_bound = int(round(hue_shift / 2))
        self.sat_bound = int(round(sat_shift * 255))
        self.val_bound = int(round(val_shift * 255))

    def __call__(self, clip, target):
        if clip.jitter is not None:
            raise NotImplementedError("two ColorJitter stages on one clip")
        hue_s = random.randint(-self.hue_bound, self.hue_bound)
        sat_s = random.randint(-self.sat_bound, self.sat_bound)
        val_s = random.randint(-self.val_bound, self.val_bound)
        clip.jitter = (hue_s, sat_s, val_s)
        return clip, target


class ToTensor:
    def __call__(self, imgs, target):
        return imgs, target            # the uint8 -> fp32 step is part of the normalisation table


class Normalize:
    """video_transforms.py:308-322: per-channel (x/255 - mean)/std on the pixels (deferred); boxes -> cxcywh / (w, h, w, h)."""

    def __init__(self, mean, std):
        self.mean, self.std = tuple(mean), tuple(std)

    def __call__(self, images, target=None):
        images.norm = (self.mean, self.std)
        if target is None:
            return images, None
        target = target.copy()
        h, w = images.height, images.width
        if "boxes" in target:
            boxes = box_xyxy_to_cxcywh(target["boxes"][:, 1:])
            boxes = boxes / torch.tensor([w, h, w, h], dtype=torch.float32)
            target["boxes"][:, 1:] = boxes
        return images, target


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, images, target):
        for t in self.transforms:
            images, target = t(images, target)
        return images, target


def make_transforms(image_set, cfg):
    """datasets/ava_frame.py:158-183 / jhmdb_frame.py:229-256 (identical pipelines)."""
    normalize = Compose([ToTensor(), Normalize(MEAN, STD)])
    if image_set == "train":
        return Compose([RandomHorizontalFlip(), RandomSizeCrop_Custom(cfg.CONFIG.DATA.IMG_SIZE), ColorJitter(), normalize])
    if image_set in ("val", "visual"):
        return Compose([Resize_Custom(cfg.CONFIG.DATA.IMG_SIZE), normalize])
    raise ValueError(f"unknown {image_set}")


# ----------------------------------------------------------------------------------------------------------------------
# collate + device pre-pass
# ----------------------------------------------------------------------------------------------------------------------
_tables = {}


def _device_tables(device, norm):
    key = (str(device), norm)
    if key not in _tables:
        _tables[key] = (torch.from_numpy(normalize_lut(*norm)).to(device), torch.from_numpy(hsv_tables()).to(device))
    return _tables[key]


_coeffs = {}


def _device_coeffs(device, H, W, oh, ow):
    key = (str(device), H, W, oh, ow)
    if key not in _coeffs:
        bh, kh = resize_coeffs(W, ow)
        bv, kv = resize_coeffs(H, oh)
        y0, y1 = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
        if ow != W:                              # the vertical pass reads the horizontally resized rows [y0, y1)
            bv = bv.copy()
            bv[:, 0] -= y0
        else:
            y0, y1 = 0, H
        dev = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in (bh, kh, bv, kv)]
        _coeffs[key] = (dev, kh.shape[1], kv.shape[1], y0, y1 - y0)
    return _coeffs[key]


class ClipBatch:
    """What ``collate_fn`` returns in place of the reference's ``NestedTensor``: the samples' :class:`FrameClip` s, still on the host.
    ``.to(device)`` runs the HIP pre-pass and returns the ``NestedTensor`` (fp32 [N,3,T,Hmax,Wmax] + bool mask [N,Hmax,Wmax])."""

    def __init__(self, clips):
        self.clips = list(clips)

    def __len__(self):
        return len(self.clips)

    def pin_memory(self):
        """``DataLoader(pin_memory=True)`` calls this on a custom batch object (in its pin thread, off the training loop): the uint8
        frames move to page-locked memory once, so ``.to(device)`` is an asynchronous copy instead of a staged, host-blocking one --
        what the reference gets for its fp32 NestedTensor from the same flag (datasets/ava_frame.py:279)."""
        for c in self.clips:
            if c.pinned is None:
                c.pinned = torch.from_numpy(c.frames).pin_memory()
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("ClipBatch.to: the clip pre-pass runs on the GPU only (HIP kernels, no CPU fallback)")
        clips = self.clips
        T = len(clips[0])
        norm = clips[0].norm or (MEAN, STD)
        for c in clips:
            if len(c) != T:
                raise ValueError("clips of one batch must have the same number of frames")
            if (c.norm or (MEAN, STD)) != norm:
                raise ValueError("clips of one batch must share the normalisation constants")
        # one staging buffer with every clip's frames at their working resolution
        offs, total = [], 0
        for c in clips:
            H, W = c.frame_hw
            offs.append(total)
            total += (T * H * W * 3 + 255) // 256 * 256
        staging = torch.empty(total, dtype=torch.uint8, device=device)
        for c, off in zip(clips, offs):
            H, W = c.frame_hw
            nbytes = T * H * W * 3
            host = c.pinned if c.pinned is not None else torch.from_numpy(c.frames)
            if c.resize_hw is None:
                staging[off:off + nbytes].view(T, H, W, 3).copy_(host, non_blocking=True)
            else:
                H0, W0 = (int(v) for v in c.frames.shape[1:3])
                src = host.to(device, non_blocking=True)
                (bh, kh, bv, kv), ksh, ksv, y0, rows = _device_coeffs(device, H0, W0, H, W)
                tmp = torch.empty(T * rows * W * 3, dtype=torch.uint8, device=device) if (W != W0 and H != H0) else None
                lib.call("tuber_frames_resize", src, tmp, staging[off:], T, H0, W0, H, W, bh, kh, ksh, bv, kv, ksv, y0, rows)
        desc = np.zeros(len(clips), _DESC)
        for i, (c, off) in enumerate(zip(clips, offs)):
            desc[i] = c.descriptor(off)
        if lib.query("tuber_clip_desc_bytes") != _DESC.itemsize:
            raise RuntimeError("TuberClipDesc layout drift between input_pipeline.py and libtuber_hip.so")
        ddesc = torch.from_numpy(desc.view(np.uint8).copy()).to(device, non_blocking=True)
        Hmax = max(c.height for c in clips)
        Wmax = max(c.width for c in clips)
        lut, hsv = _device_tables(device, norm)
        out = torch.empty(len(clips), 3, T, Hmax, Wmax, dtype=torch.float32, device=device)
        mask = torch.empty(len(clips), Hmax, Wmax, dtype=torch.bool, device=device)
        lib.call("tuber_clip_prepare", staging, ddesc, lut, hsv, out, mask, len(clips), T, Hmax, Wmax)
        return NestedTensor(out, mask)


def collate_fn(batch):
    """``utils/misc.py:279-282`` with the padding deferred to the device: [(FrameClip, target), ...] -> (ClipBatch, (targets...))."""
    batch = list(zip(*batch))
    batch[0] = ClipBatch(batch[0])
    return tuple(batch)
