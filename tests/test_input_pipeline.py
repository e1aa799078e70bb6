"""Clip input pipeline.

CPU part: the numpy oracle against Pillow itself and against the vectors generated from the reference's own transform pipelines
(oracle/gen_input_golden.py -> tests/golden/input_pipeline.npz), and the host logic of tubelet_transformer_amd.input_pipeline
(random draw order, flip/crop composition, box bookkeeping, resampling tables).
GPU part (-m gpu): the HIP pre-pass (tuber_frames_resize, tuber_clip_prepare) through the C ABI, bit-exact against the oracle / vectors.
"""
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import input_pipeline_ref as R                      # noqa: E402
from tubelet_transformer_amd import input_pipeline as P         # noqa: E402
from tubelet_transformer_amd.config import get_cfg_defaults     # noqa: E402


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "input_pipeline.npz"))


def _cfg(size):
    cfg = get_cfg_defaults()
    cfg.CONFIG.DATA.IMG_SIZE = size
    return cfg


def _replay(gold, k):
    """run THIS repo's transforms on sample k of the golden file with the reference's seed -> (FrameClip, target)."""
    mode, seed, nh, nw = (int(v) for v in gold["sample_meta"][k])
    clip = P.FrameClip(gold["s%d_frames" % k]).resize((nw, nh))
    target = {"boxes": torch.from_numpy(gold["s%d_in_boxes" % k].copy()), "raw_boxes": torch.from_numpy(gold["s%d_in_raw_boxes" % k].copy()),
              "labels": torch.from_numpy(gold["s%d_in_labels" % k].copy()), "orig_size": torch.as_tensor([nh, nw]),
              "size": torch.as_tensor([nh, nw])}
    tf = P.make_transforms("train" if mode == 0 else "val", _cfg(40))
    random.seed(seed)
    return tf(clip, target)


def _oracle_clip(clip):
    _, H, W, y1, x1, h, w, flip, jit, hue, sat, val, _ = clip.descriptor(0)
    return R.prepare_clip(clip.frames, resize_hw=clip.resize_hw, flip=bool(flip), crop=(y1, x1, h, w), jitter=(hue, sat, val) if jit else None)


# ------------------------------------------------------------------ CPU ------------------------------------------------------------------
def test_input_pipeline_resize_matches_pillow(gold):
    """oracle.pil_resize == PIL.Image.resize: committed vectors always, Pillow itself when importable (fresh random sizes)."""
    for k, (H, W, oh, ow) in enumerate(gold["resize_cases"]):
        got = R.pil_resize(gold["resize_in_%d" % k], int(oh), int(ow))
        assert np.array_equal(got, gold["resize_out_%d" % k]), (H, W, oh, ow)
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    for H, W, oh, ow in [(37, 91, 64, 64), (120, 160, 96, 128), (33, 20, 33, 47), (200, 100, 31, 100)]:
        a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        assert np.array_equal(R.pil_resize(a, oh, ow), np.asarray(Image.fromarray(a).resize((ow, oh))))


def test_input_pipeline_host_tables_match_oracle():
    for n_in, n_out in [(80, 64), (64, 80), (91, 17), (48, 48), (1080, 288), (5, 9)]:
        b0, k0 = R.resize_coeffs(n_in, n_out)
        b1, k1 = P.resize_coeffs(n_in, n_out)
        assert np.array_equal(b0, b1) and np.array_equal(k0, k1), (n_in, n_out)
    assert np.array_equal(R.normalize_lut(), P.normalize_lut())
    assert np.array_equal(np.stack(R.hsv_tables()), P.hsv_tables())
    assert P._DESC.itemsize == 56                                          # long long + 12 ints, as TuberClipDesc in include/tuber_hip.h


def test_input_pipeline_hsv_invariants():
    """the OpenCV restatement is unpinned (cv2 absent): check what must hold for ANY correct 8-bit HSV pair."""
    grey = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)[None]
    hsv = R.rgb2hsv_u8(grey)
    assert (hsv[..., 0] == 0).all() and (hsv[..., 1] == 0).all() and np.array_equal(hsv[..., 2], grey[..., 0])
    assert np.array_equal(R.hsv2rgb_u8(hsv), grey)
    prim = np.array([[[255, 0, 0], [255, 255, 0], [0, 255, 0], [0, 255, 255], [0, 0, 255], [255, 0, 255]]], np.uint8)
    assert np.array_equal(R.rgb2hsv_u8(prim)[0, :, 0], [0, 30, 60, 90, 120, 150])
    assert np.array_equal(R.hsv2rgb_u8(R.rgb2hsv_u8(prim)), prim)
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (2, 40, 40, 3), dtype=np.uint8)
    assert np.abs(R.color_jitter(a, 0, 0, 0).astype(int) - a).max() <= 6          # 8-bit hue quantisation
    assert np.array_equal(R.color_jitter(a, 180 - 180, 0, 0), R.color_jitter(a, 0, 0, 0))
    bright = R.color_jitter(a, 0, 0, 26).astype(int).max(-1)
    assert (bright >= a.astype(int).max(-1)).all()                                    # V = max(r,g,b) only grows


def test_input_pipeline_transforms_match_reference(gold):
    """same seed -> same draws, geometry, pixels and boxes as the reference's make_transforms('train'|'val') pipelines."""
    for k in range(len(gold["sample_meta"])):
        clip, tgt = _replay(gold, k)
        for f in ("boxes", "raw_boxes", "labels", "size", "area"):
            assert np.array_equal(tgt[f].numpy(), gold["s%d_out_%s" % (k, f)]), (k, f)
        got = _oracle_clip(clip)
        assert got.shape == gold["s%d_clip" % k].shape and np.array_equal(got, gold["s%d_clip" % k]), k
    # the training samples must actually exercise flip, crop offsets and jitter
    plans = [_replay(gold, k)[0] for k in range(4)]
    assert any(p.sx < 0 for p in plans) and any(p.sx > 0 for p in plans)
    assert all(p.jitter is not None for p in plans) and len({p.jitter for p in plans}) > 1


def test_input_pipeline_flip_crop_compose_in_any_order():
    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, (2, 20, 30, 3), dtype=np.uint8)
    clip = P.FrameClip(frames)
    t = {"labels": torch.zeros(0)}
    P.crop(clip, t, (2, 3, 15, 20)); P.hflip(clip, t); P.crop(clip, t, (1, 4, 10, 9)); P.hflip(clip, t)
    want = frames[:, 2:17, 3:23][:, :, ::-1][:, 1:11, 4:13][:, :, ::-1]
    lut = R.normalize_lut()
    assert np.array_equal(_oracle_clip(clip), np.stack([lut[c][want[..., c]] for c in range(3)]))
    with pytest.raises(ValueError):
        P.crop(clip, t, (0, 0, 11, 9))
    with pytest.raises(NotImplementedError):
        clip.resize((10, 10))


def test_input_pipeline_needs_the_gpu():
    clip = P.FrameClip(np.zeros((2, 8, 8, 3), np.uint8))
    batch, _ = P.collate_fn([(clip, {}), (clip, {})])
    with pytest.raises(RuntimeError):
        batch.to("cpu")


# ------------------------------------------------------------------ GPU ------------------------------------------------------------------
def _resize_gpu(dev, frames, oh, ow):
    T, H, W = frames.shape[:3]
    clip = P.FrameClip(frames).resize((ow, oh))
    out = P.ClipBatch([clip]).to(dev)
    return clip, out


@pytest.mark.gpu
def test_gpu_resize_bit_exact(dev, gold):
    from tubelet_transformer_amd import lib
    cases = [(gold["resize_in_%d" % k], int(oh), int(ow), gold["resize_out_%d" % k]) for k, (_, _, oh, ow) in enumerate(gold["resize_cases"])]
    rng = np.random.default_rng(3)
    for H, W, oh, ow in [(360, 480, 288, 384), (240, 320, 288, 384), (96, 130, 96, 77), (130, 96, 61, 96), (31, 17, 90, 50), (8, 4100, 8, 2050),
                         (50, 64, 40, 48), (33, 47, 21, 64)]:
        a = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
        cases.append((a, oh, ow, R.pil_resize(a, oh, ow)))
    for a, oh, ow, want in cases:
        T, H, W = a.shape[:3]
        (bh, kh, bv, kv), ksh, ksv, y0, rows = P._device_coeffs(dev, H, W, oh, ow)
        src = torch.from_numpy(a).to(dev)
        dst = torch.empty(T, oh, ow, 3, dtype=torch.uint8, device=dev)
        tmp = torch.empty(T * rows * ow * 3, dtype=torch.uint8, device=dev)
        lib.call("tuber_frames_resize", src, tmp, dst, T, H, W, oh, ow, bh, kh, ksh, bv, kv, ksv, y0, rows)
        assert np.array_equal(dst.cpu().numpy(), want), (H, W, oh, ow)


@pytest.mark.gpu
def test_gpu_clip_batch_matches_reference_vectors(dev, gold):
    """the six reference-generated samples, collated raggedly in one batch: fp32 pixels, zero padding and mask, bit for bit."""
    replays = [_replay(gold, k) for k in range(len(gold["sample_meta"]))]
    batch, targets = P.collate_fn(replays)
    nt = batch.to(dev)
    want, wmask = R.collate([gold["s%d_clip" % k] for k in range(len(replays))])
    assert nt.tensors.shape == want.shape and nt.mask.dtype == torch.bool
    assert np.array_equal(nt.tensors.cpu().numpy(), want)
    assert np.array_equal(nt.mask.cpu().numpy(), wmask)
    assert len(targets) == len(replays)


@pytest.mark.gpu
@pytest.mark.parametrize("wcrop", [340, 339, 333])     # 340: float4 stores; others: scalar tail path
def test_gpu_clip_prepare_full_size(dev, wcrop):
    """BASELINE-sized clips (32 x 256 x 340 after the crop) with resize, flip, jitter on one clip and none on the other."""
    rng = np.random.default_rng(wcrop)
    f0 = rng.integers(0, 256, (32, 270, 360, 3), dtype=np.uint8)
    f1 = rng.integers(0, 256, (32, 288, 384, 3), dtype=np.uint8)
    c0 = P.FrameClip(f0).resize((384, 288))
    c1 = P.FrameClip(f1)
    t = {"labels": torch.zeros(0)}
    P.hflip(c0, t); P.crop(c0, t, (17, 23, 256, wcrop)); c0.jitter = (-7, 19, -26)
    P.crop(c1, t, (0, 5, 250, wcrop - 9))
    nt = P.ClipBatch([c0, c1]).to(dev)
    want, wmask = R.collate([_oracle_clip(c0), _oracle_clip(c1)])
    got = nt.tensors.cpu().numpy()
    assert got.shape == want.shape == (2, 3, 32, 256, wcrop)
    assert np.array_equal(got, want)
    assert np.array_equal(nt.mask.cpu().numpy(), wmask)


@pytest.mark.gpu
def test_gpu_jitter_exhaustive_hue_sat_val_extremes(dev):
    """every (r,g,b) on a 52-level grid through the device HSV jitter at the bounds of the three shifts vs the oracle."""
    lv = np.arange(0, 256, 5, dtype=np.uint8)
    grid = np.stack(np.meshgrid(lv, lv, lv, indexing="ij"), -1).reshape(1, 52, 52 * 52, 3)
    for jit in [(0, 0, 0), (10, 26, 26), (-10, -26, -26), (3, -26, 26), (-9, 11, 0)]:
        c = P.FrameClip(grid.copy())
        c.jitter = jit
        nt = P.ClipBatch([c]).to(dev)
        assert np.array_equal(nt.tensors.cpu().numpy()[0], _oracle_clip(c)), jit


@pytest.mark.gpu
def test_gpu_training_epoch_from_uint8_frames(dev):
    """rows N3 + (a) end to end: decoded uint8 frames -> reference-named transforms -> collate_fn -> train_tuber_detection, which
    calls samples.to(device) (the HIP pre-pass) and steps the model; then the eval transforms through validate-style forward."""
    from tubelet_transformer_amd import synth
    from tubelet_transformer_amd.config import load_cfg
    from tubelet_transformer_amd.training import build_optimizer, train_tuber_detection
    from tubelet_transformer_amd.tuber import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_cfg(os.path.join(root, "configuration", "TubeR_CSN50_AVA21.yaml"))
    cfg.CONFIG.DATA.IMG_SIZE = 64
    model, criterion, _ = build_model(cfg)
    synth.load_name_hashed(model)
    model.to(dev)
    criterion.to(dev)
    rng = np.random.default_rng(0)
    tf = P.make_transforms("train", cfg)
    random.seed(3)
    loader = []
    for it in range(2):
        samples = []
        for b in range(2):
            nh, nw = 72, (96 if b == 0 else 90)                   # ragged widths -> padding + mask inside the batch
            clip = P.FrameClip(rng.integers(0, 256, (32, 80, 110, 3), dtype=np.uint8)).resize((nw, nh))
            boxes = torch.tensor([[16.0, 10, 8, 60, 60], [16.0, 30, 20, 85, 70]])
            labels = torch.zeros(2, 80); labels[:, 11] = 1; labels[1, 40] = 1
            target = {"image_id": ["v_%d" % it, 16], "boxes": boxes, "raw_boxes": torch.nn.functional.pad(boxes, (1, 0, 0, 0), value=b),
                      "labels": labels, "orig_size": torch.as_tensor([nh, nw]), "size": torch.as_tensor([nh, nw])}
            samples.append(tf(clip, target))
        loader.append(P.collate_fn(samples))
    nt = loader[0][0].to(dev)
    assert nt.tensors.shape[:3] == (2, 3, 32) and nt.tensors.shape[3] == 64 and bool(nt.mask.any()) and not bool(nt.mask.all())
    opt = build_optimizer(model, cfg)
    w0 = model.class_fc.weight.detach().clone()
    loss = train_tuber_detection(cfg, model, criterion, loader, opt, epoch=0, max_norm=cfg.CONFIG.LOSS_COFS.CLIPS_MAX_NORM, print_freq=100)
    assert torch.isfinite(loss) and not torch.equal(w0, model.class_fc.weight.detach())
