"""CPU ORACLE for the clip input pipeline -- TEST INFRASTRUCTURE ONLY.

numpy restatement of what the reference does to a clip between the JPEG decoder and ``model(samples)``:

  ``datasets/ava_frame.py:133-152``   ``loadvideo``: every frame ``PIL.Image.resize((nw, nh))`` (Pillow default filter: bicubic)
  ``datasets/ava_frame.py:158-176``   ``make_transforms``: train = flip -> RandomSizeCrop_Custom -> ColorJitter -> ToTensor+Normalize;
                                      val = Resize_Custom (a centre "fake crop") -> ToTensor+Normalize
  ``datasets/video_transforms.py:20-66,69-85,184-208,210-227,308-322,333-369``  the transforms themselves
  ``datasets/ava_frame.py:71-74``     stack to (3,T,H,W)
  ``utils/misc.py:279-282,367-425``   ``collate_fn`` -> ``nested_tensor_from_tensor_list`` (zero pad to the batch max, bool mask)

Only ``tests/`` may import this file; the shipped path (``tubelet_transformer_amd/input_pipeline.py``) runs HIP kernels only.

Pinning, leg by leg:
  * resize: the arithmetic lives in Pillow (``src/libImaging/Resample.c``; the reference pins no version, 12.2.0 is in this image).
    ``pil_resize`` restates its 8-bit two-pass fixed-point bicubic and is pinned bit-exactly against ``PIL.Image.resize`` itself
    (``tests/test_cpu.py::test_input_pipeline_resize_matches_pillow`` + ``tests/golden/input_pipeline.npz``).
  * flip / crop / box bookkeeping: pinned against the reference's own ``hflip`` / ``crop`` / ``Normalize`` / ``RandomSizeCrop_Custom`` /
    ``Resize_Custom`` imported by ``oracle/gen_input_golden.py`` (targets and random draw order; the image side of those functions is
    ``PIL.Image.crop`` / ``transpose`` = pure indexing).
  * ToTensor + Normalize: torchvision semantics (``u8 / 255`` in fp32, then ``(x - mean) / std`` in fp32).  torchvision is not in this
    image: PARITY UNPINNED by import, restated from its documented behaviour.
  * ColorJitter: the arithmetic lives in OpenCV (``cv2.cvtColor`` 8-bit ``COLOR_RGB2HSV`` / ``COLOR_HSV2RGB``,
    ``modules/imgproc/src/color_hsv.simd.hpp``; no version pinned by the reference, cv2 is not in this image): PARITY UNPINNED --
    ``rgb2hsv_u8`` / ``hsv2rgb_u8`` restate the published scalar algorithm (fixed-point 12-bit division tables for RGB->HSV, fp32
    sector formula + round-half-even for HSV->RGB) and are checked only through their invariants (grey pixels, primary colours,
    identity jitter round trip within 8-bit quantisation).
"""
import numpy as np

MEAN = (0.485, 0.456, 0.406)    # datasets/ava_frame.py:161
STD = (0.229, 0.224, 0.225)

# --------------------------------------------------------------------------------------------------------------
# Pillow 8-bit bicubic resize (Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc)
# --------------------------------------------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resize_coeffs(in_size, out_size):
    """-> (bounds [out,2] int32 = (first tap, tap count), kk [out,ksize] int32 fixed-point weights)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)], np.float64)
        ww = 0.0
        for v in w:          # same left-to-right double sum as the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        q = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))
        kk[xx, :xmax] = np.trunc(q).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample(img, bounds, kk, axis):
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((len(bounds),) + src.shape[1:], np.uint8)
    for i, (lo, n) in enumerate(bounds):
        acc = (src[lo:lo + n] * kk[i, :n].reshape((-1,) + (1,) * (src.ndim - 1))).sum(0) + (1 << (PRECISION_BITS - 1))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def pil_resize(img, oh, ow):
    """``Image.fromarray(img).resize((ow, oh))`` for uint8 [H,W,3] (or [T,H,W,3]: every frame alike)."""
    if img.ndim == 4:
        return np.stack([pil_resize(f, oh, ow) for f in img])
    H, W = img.shape[:2]
    if (oh, ow) == (H, W):
        return img.copy()
    bh, kh = resize_coeffs(W, ow)
    bv, kv = resize_coeffs(H, oh)
    out = img
    if ow != W:                              # horizontal pass, only over the rows the vertical pass will read
        y0, y1 = bv[0, 0], bv[-1, 0] + bv[-1, 1]
        out = _resample(out[y0:y1], bh, kh, 1)
        bv = bv.copy()
        bv[:, 0] -= y0
    if oh != H:
        out = _resample(out, bv, kv, 0)
    return out


# --------------------------------------------------------------------------------------------------------------
# OpenCV 8-bit HSV (hue range 180)
# --------------------------------------------------------------------------------------------------------------
HSV_SHIFT = 12


def hsv_tables():
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256, np.int32)
    hdiv = np.zeros(256, np.int32)
    sdiv[1:] = np.rint((255 << HSV_SHIFT) / (1.0 * i)).astype(np.int32)      # saturate_cast<int>(double) = round half even
    hdiv[1:] = np.rint((180 << HSV_SHIFT) / (6.0 * i)).astype(np.int32)
    return sdiv, hdiv


def rgb2hsv_u8(rgb):
    sdiv, hdiv = hsv_tables()
    r, g, b = (rgb[..., k].astype(np.int32) for k in range(3))
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s = (diff * sdiv[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + (~vg & (r - g + 4 * diff))))
    h = (h * hdiv[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s & 255, v], -1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])   # (b, g, r) picks per sector


def hsv2rgb_u8(hsv):
    f = np.float32
    h = hsv[..., 0].astype(f) * f(6.0 / 180.0)
    s = hsv[..., 1].astype(f) * f(1.0 / 255.0)
    v = hsv[..., 2].astype(f) * f(1.0 / 255.0)
    h = np.where(h >= f(6), h - f(6), h).astype(f)
    sector = np.floor(h).astype(np.int32)
    h = (h - sector.astype(f)).astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    h = np.where(bad, f(0), h).astype(f)
    one = f(1)
    tab = np.stack([v, v * (one - s), v * (one - s * h), v * (one - s * (one - h))], -1).astype(f)
    pick = _SECTOR[sector]                                         # [...,3] indices for b,g,r
    bgr = np.take_along_axis(tab, pick, -1)
    bgr = np.where((hsv[..., 1] == 0)[..., None], v[..., None], bgr).astype(f)
    out = np.clip(np.rint(bgr * f(255.0)), 0, 255).astype(np.uint8)   # saturate_cast<uchar>(float): round half even
    return out[..., ::-1]                                          # -> r,g,b


def color_jitter(clip, hue_s, sat_s, val_s):
    """``ColorJitter.__call__`` with the three random draws given (video_transforms.py:333-369)."""
    hsv = rgb2hsv_u8(clip).astype(np.int32)
    hsv[..., 0] = (hsv[..., 0] + hue_s + 180) % 180
    hsv[..., 1] = np.clip(hsv[..., 1] + sat_s, 0, 255)
    hsv[..., 2] = np.clip(hsv[..., 2] + val_s, 0, 255)
    return hsv2rgb_u8(hsv.astype(np.uint8))


# --------------------------------------------------------------------------------------------------------------
# clip -> normalised fp32 tensor, batch collate
# --------------------------------------------------------------------------------------------------------------
def normalize_lut():
    """[3,256] fp32: ToTensor (u8/255) then Normalize ((x-mean)/std), both in fp32 like torchvision."""
    u = np.arange(256, dtype=np.float32) / np.float32(255)
    return np.stack([(u - np.float32(m)) / np.float32(s) for m, s in zip(MEAN, STD)]).astype(np.float32)


def prepare_clip(frames, resize_hw=None, flip=False, crop=None, jitter=None):
    """frames uint8 [T,H0,W0,3] -> fp32 [3,T,h,w].  Order as the reference: resize, flip, crop (y1,x1,h,w), jitter, normalise."""
    x = frames
    if resize_hw is not None:
        x = pil_resize(x, *resize_hw)
    if flip:
        x = x[:, :, ::-1]
    if crop is not None:
        y1, x1, h, w = crop
        x = x[:, y1:y1 + h, x1:x1 + w]
    if jitter is not None:
        x = color_jitter(x, *jitter)
    lut = normalize_lut()
    out = np.stack([lut[c][x[..., c]] for c in range(3)])          # [3,T,h,w]
    return out


def collate(clips):
    """``nested_tensor_from_tensor_list`` on (3,T,h,w) clips: zero pad bottom/right, mask True on padding."""
    T = clips[0].shape[1]
    H = max(c.shape[2] for c in clips)
    W = max(c.shape[3] for c in clips)
    out = np.zeros((len(clips), 3, T, H, W), np.float32)
    mask = np.ones((len(clips), H, W), bool)
    for i, c in enumerate(clips):
        out[i, :, :, :c.shape[2], :c.shape[3]] = c
        mask[i, :c.shape[2], :c.shape[3]] = False
    return out, mask
