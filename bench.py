#!/usr/bin/env python
"""TubeR training-step throughput on MI355X (the headline metric of BASELINE.json).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...            (no launcher: bench.py starts its own N ranks, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one full optimisation step of TubeR_CSN152_AVA21 (forward, Hungarian-matched criterion, backward, global-norm
clip 0.1, AdamW) on 2 synthetic clips 3x32x256x340 per GPU, model in train() mode with dropout ON, inputs resident in HBM.
Rank 0 prints ONE JSON line: whole-job clips/s, plus
  * "roofline": the dominant kernel family of the step (picked by a HIP-event pre-pass over every launch), its
    ALGORITHMIC bytes per launch / average launch duration measured with HIP events INSIDE the timed steps, vs the 8 TB/s HBM peak;
  * "cpu_baseline": the CPU oracle (stock PyTorch CPU ops, fp32, same graph) timed on this box's host cores on ONE clip
    of the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts: RCCL needs it before the runtime starts

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def alg_cost(name, a):
    """(key, algorithmic bytes, flops) of one launch from its C-ABI arguments (DESIGN.md 'algorithmic bytes')."""
    from tubelet_transformer_amd import lib
    if name == "tuber_gemm_nt":
        M, N, K = a[6], a[7], a[8]
        amode, epi, out_f32 = a[9], a[21], a[26]
        cfg = lib.query("tuber_gemm_nt_cfg", M, N, K)
        by = 2 * (M * K + N * K) + (4 if out_f32 else 2) * M * N
        if a[23] is not None:
            by += 2 * M * N          # residual read
        if epi == 2:
            by += 2 * M * N          # mask source read
        tile, occ = {0: ("128,128,2,2,2", 2), 2: ("64,64,2,2,4", 4), 7: ("64,128,1,4,2", 3), 12: ("64,64,2,2,4", 3), 13: ("64,64,2,2,2", 4),
                     17: ("64,128,1,4,2", 4)}[cfg]
        if amode == 0 and not a[12] and M >= 8192 and N % 64 == 0 and not out_f32 and epi in (0, 1, 2):
            tile, occ = "96,64,2,2,2", 4     # round 6: plain-A shapes with >= 8 192 rows run on 96 x 64 tiles (gemm.hip: nt_use_96)
        return "gemm_nt_kernel<%s,%d,%d,%d>" % (tile, amode, epi, occ), by, 2 * M * N * K
    if name in ("tuber_gemm_nt_join", "tuber_gemm_nt_join_mask"):
        M, N, K = a[6], a[7], a[8]
        cfg = lib.query("tuber_gemm_nt_cfg", M, N, K)
        tile, occ = {0: ("64,128,1,4,2", 3), 7: ("64,128,1,4,2", 3), 13: ("64,64,2,2,2", 4)}.get(cfg, ("64,64,2,2,2", 4))
        masked = name.endswith("_mask")      # round 6: the ReLU mask of y as a bit field (1 bit instead of 2 bytes per element), epilogue EPI_JOIN_M = 7
        by = 2 * (M * K + N * K + M * N) + 2 * M * N * (2 if a[9] is not None else 1) + (M * N // 8 if masked else 2 * M * N)      # + residual, statistics operand c4, mask
        return "gemm_nt_kernel<%s,0,%d,%d>" % (tile, 7 if masked else 3, occ), by, 2 * M * N * K
    if name == "tuber_gemm_tn":
        M, N, K = a[7], a[8], a[9]
        ldg, lda, gmode = a[1], a[3], a[22] is not None
        if not gmode and not ((N | K | ldg | lda) & 7):      # the LDS-transpose-read kernels (gemm.hip: gemm_tn2_kernel / gemm_tn3_kernel)
            return "gemm_tn%d_kernel<%d>" % (3 if lib.query("tuber_gemm_tn_tile", M, N, K) == 128 else 2, a[10]), 2 * M * (N + K) + 4 * N * K, 2 * M * N * K
        T = 128 if ((N + 127) // 128) * ((K + 127) // 128) >= 128 else 64
        return "gemm_tn_kernel<%d,%d,%d>" % (a[10], T, 1 if gmode else 0), 2 * M * (N + K) + 4 * N * K, 2 * M * N * K
    if name == "tuber_gemm_tn_group":
        by = sum(2 * e.M * (e.N + e.K) + 4 * e.N * e.K for e in a[0])
        tiles = {lib.query("tuber_gemm_tn_tile", e.M, e.N, e.K) for e in a[0]}
        key = "gemm_tn3_group_kernel" if tiles == {128} else "gemm_tn2_group_kernel" if tiles == {64} else "gemm_tn2+tn3_group_kernels"
        return key, by, sum(2 * e.M * e.N * e.K for e in a[0])
    if name in ("tuber_dwconv_fwd", "tuber_dwconv_bwd_data", "tuber_dwconv_bwd_weight"):
        off = {"tuber_dwconv_fwd": 7, "tuber_dwconv_bwd_data": 8, "tuber_dwconv_bwd_weight": 7}[name]
        N, Ti, Hi, Wi, To, Ho, Wo, C, st, ss = a[off:off + 10]
        by = 2 * C * N * (Ti * Hi * Wi + To * Ho * Wo)
        if name == "tuber_dwconv_bwd_data":
            by += 2 * C * N * Ti * Hi * Wi                       # reads x for the relu/bn mask, writes dz
        key = {"tuber_dwconv_fwd": "dwconv_fwd_kernel<%d>" % ss, "tuber_dwconv_bwd_data": "dwconv_bwd_data_kernel<%d>" % ss,
               "tuber_dwconv_bwd_weight": "dwconv_bwd_weight_kernel<%d>" % ss}[name]
        return key, by, 2 * 27 * C * N * To * Ho * Wo
    if name == "tuber_bn_bwd_fa":
        return "bn_bwd_fa_kernel", 2 * 3 * a[13] * a[3], 0
    if name == "tuber_dwconv_tile_bwd_data_bn":
        N, T, H, W, C = a[18:23]
        return "dwconv_tile_kernel<1,true>", 2 * C * N * T * H * W * 4, 2 * 27 * C * N * T * H * W
    if name == "tuber_dwconv_tile_bwd_both_bn":
        # backward of a depthwise conv = read dy (dzu), read the saved conv output xu (the BatchNorm-backward fold and
        # the ReLU mask need it -- it stands for "dy" of the un-fused conv), read x, write dx = FOUR tensor passes.  Whatever the kernel
        # re-reads on top of that (r03: two concatenated grids, 7 passes) is traffic, not algorithmic bytes.
        N, T, H, W, C = a[19:24]
        return "dwconv_tile_bwd_both_kernel", 2 * C * N * T * H * W * 4, 2 * 2 * 27 * C * N * T * H * W
    if name == "tuber_dwconv_tile_bwd_weight_bn":
        N, T, H, W, C = a[15:20]
        return "dwconv_tile_kernel<2,true>", 2 * C * N * T * H * W * 3, 2 * 27 * C * N * T * H * W
    if name == "tuber_dwconv_tile_fwd_bn":       # the forward conv that also finalises bn1 (kernel name dwconv_tile_fwd_fin_kernel)
        N, T, H, W, C = a[20:25]
        return "dwconv_tile_fwd_fin_kernel", 2 * C * N * T * H * W * 2, 2 * 27 * C * N * T * H * W
    if name in ("tuber_dwconv_tile_fwd", "tuber_dwconv_tile_bwd_data", "tuber_dwconv_tile_bwd_weight"):
        off = 8 if name == "tuber_dwconv_tile_bwd_data" else 7
        N, T, H, W, C = a[off:off + 5]
        el = C * N * T * H * W
        mode = {"tuber_dwconv_tile_fwd": 0, "tuber_dwconv_tile_bwd_data": 1, "tuber_dwconv_tile_bwd_weight": 2}[name]
        return "dwconv_tile_kernel<%d>" % mode, 2 * el * (3 if mode == 1 else 2), 2 * 27 * el
    if name == "tuber_conv4_bwd_fused":          # reads dz, c4 [M,256] and c3 [M,64], writes dz3 [M,64]; conv4 data + weight gradient
        M = a[14]
        return "conv4_bwd_kernel<%s>" % ("true" if a[8] is None else "false"), 2 * M * (256 + 256 + 64 + 64), 2 * 2 * M * 256 * 64
    if name == "tuber_entry_conv_fwd":           # reads x [M,64], writes c1 [M,64] and cd [M,256]
        M = a[11]
        return "entry_conv_kernel", 2 * M * (64 + 64 + 256), 2 * M * 64 * 320
    if name == "tuber_stem_conv_bwd_weight_bn":  # reads the clip, dz0 and c0 [Mo,64]
        B, T, H, W = a[9:13]
        Mo = B * T * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        return "stem_conv_bwd_w_kernel", 4 * B * 3 * T * H * W + 2 * 2 * Mo * 64, 2 * Mo * 64 * 441
    if name == "tuber_blockout_conv1_fwd":       # reads c4 and the shortcut, writes y [M,256] and the next conv1 output [M,pn]
        M, pn = a[12], a[13]
        return "blockout_conv1_kernel<%d>" % pn, 2 * M * (3 * 256 + pn), 2 * M * 256 * pn
    if name == "tuber_conv1_bwd_fused":          # reads dz1, c1 [M,64], x [M,256] (+ residual gradient, + lower c4, + lower projection output), writes [M,256]
        M = a[16]
        wide = 2 + (a[7] is not None) + (a[9] is not None) + (a[10] is not None)
        return "conv1_bwd_kernel<%d>" % (2 if a[10] is not None else 1 if a[9] is not None else 0), 2 * M * (128 + 256 * wide), 2 * 2 * M * 256 * 64
    if name == "tuber_block_out_fwd":
        return "block_out_fwd_kernel", 2 * 3 * a[7] * a[8], 0
    if name == "tuber_block_out_bwd":
        return "block_out_bwd_kernel", 2 * (5 if a[3] is not None else 4) * a[8] * a[9], 0
    if name == "tuber_bn_bwd_apply":
        return "bn_bwd_apply_kernel", 2 * 3 * a[6] * a[7], 0
    if name == "tuber_stem_conv_fwd":
        B, T, H, W = a[5:9]
        Mo = B * T * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        return "stem_conv_fwd_kernel", 4 * B * 3 * T * H * W + 2 * Mo * 64, 2 * Mo * 64 * 441
    if name == "tuber_stem_conv_bwd_weight":
        B, T, H, W = a[5:9]
        Mo = B * T * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        return "stem_conv_bwd_w_kernel", 4 * B * 3 * T * H * W + 2 * Mo * 64, 2 * Mo * 64 * 441
    return name.replace("tuber_", "") + "*", 0, 0


def shape_of(name, a):
    if name == "tuber_gemm_nt":
        return "M%d N%d K%d amode%d epi%d" % (a[6], a[7], a[8], a[9], a[21])
    if name in ("tuber_gemm_nt_join", "tuber_gemm_nt_join_mask"):
        return "M%d N%d K%d join" % (a[6], a[7], a[8])
    if name == "tuber_gemm_tn":
        return "M%d N%d K%d amode%d" % (a[7], a[8], a[9], a[10])
    if name == "tuber_gemm_tn_group":
        return "%d x (M%d N%d K%d ..)" % (len(a[0]), a[0][0].M, a[0][0].N, a[0][0].K)
    if name in ("tuber_attn_fwd", "tuber_attn_bwd"):
        off = 10 if name == "tuber_attn_fwd" else 19
        return "B%d H%d Lq%d Lk%d" % tuple(a[off:off + 4])
    if name in ("tuber_conv4_bwd_fused", "tuber_conv1_bwd_fused"):
        return "M%d" % a[14 if name == "tuber_conv4_bwd_fused" else 16]
    if name == "tuber_entry_conv_fwd":
        return "M%d" % a[11]
    if name == "tuber_blockout_conv1_fwd":
        return "M%d pn%d" % (a[12], a[13])
    if name in ("tuber_bn_bwd_apply", "tuber_block_out_fwd", "tuber_block_out_bwd", "tuber_bn_finalize", "tuber_bn_bwd_finalize",
                "tuber_reduce_rows", "tuber_colsum", "tuber_reduce_slabs", "tuber_layernorm_fwd", "tuber_layernorm_bwd", "tuber_dropout"):
        return " ".join(str(x) for x in a if isinstance(x, int) and not isinstance(x, bool))[:44]
    if name.startswith("tuber_dwconv_tile"):
        off = {"tuber_dwconv_tile_bwd_data": 8, "tuber_dwconv_tile_bwd_data_bn": 18, "tuber_dwconv_tile_bwd_weight_bn": 15,
               "tuber_dwconv_tile_bwd_both_bn": 19}.get(name, 7)
        return "N%d %dx%dx%d C%d" % tuple(a[off:off + 5])
    if name.startswith("tuber_dwconv"):
        off = {"tuber_dwconv_fwd": 7, "tuber_dwconv_bwd_data": 8, "tuber_dwconv_bwd_weight": 7}[name]
        return "N%d in%dx%dx%d out%dx%dx%d C%d st%d ss%d" % tuple(a[off:off + 10])
    return ""


# kernel families the roofline object is ALSO given for next to the dominant one (VERDICT r02 asked for the weight-gradient GEMMs)
ALSO_TIMED = ("gemm_tn2_group_kernel", "gemm_tn3_group_kernel", "gemm_tn2+tn3_group_kernels")


class LaunchTimer:
    """HIP-event timing of kernel launches on torch's current stream (where every tuber_* launch is enqueued)."""

    def __init__(self, only=None):
        self.only, self.rec = only, []
        # an event pair around NOTHING still measures a few microseconds (event packets in the queue); calibrate that once and
        # subtract it from every launch so short kernels are not inflated (rocprofv3's kernel durations are the cross-check)
        pairs = []
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        self.overhead_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]

    def _ms(self, e0, e1):
        return max(e0.elapsed_time(e1) - self.overhead_ms, 1e-4)

    def __call__(self, name, args, launch):
        key, by, fl = alg_cost(name, args)
        if self.only is not None and key != self.only and key not in ALSO_TIMED:
            return launch(name, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = launch(name, *args)
        e1.record()
        self.rec.append((key, by, fl, e0, e1, shape_of(name, args)))
        return rc

    def by_shape(self, top=40):
        out = {}
        for key, by, fl, e0, e1, shp in self.rec:
            d = out.setdefault((key, shp), [0, 0.0, 0, 0])
            d[0] += 1
            d[1] += self._ms(e0, e1)
            d[2] += by
            d[3] += fl
        rows = sorted(out.items(), key=lambda kv: -kv[1][1])[:top]
        return ["%-40s %-44s n=%-3d %8.3f ms  %7.1f GB/s %7.1f TF/s" % (k[0], k[1], v[0], v[1], v[2] / (v[1] * 1e-3) / 1e9 if v[1] else 0,
                                                                        v[3] / (v[1] * 1e-3) / 1e12 if v[1] else 0) for k, v in rows]

    def summary(self):
        out = {}
        for key, by, fl, e0, e1, _ in self.rec:
            d = out.setdefault(key, {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0})
            d["launches"] += 1
            d["ms"] += self._ms(e0, e1)
            d["bytes"] += by
            d["flops"] += fl
        return out


def dominant_from_trace(prepass, headline):
    """The dominant kernel family = the first line of the committed rocprofv3 kernel-trace summary of this same command
    (profiles/r*_kernel_trace_stats.txt, sorted by total GPU time) that this run also launches.  HIP-event timing of an eager pass
    inflates 5-10 us launches by ~4 us each, which would rank the ~180 tiny transformer GEMMs first; the rocprof durations do not.
    Only the selection comes from the file -- the roofline numbers themselves are measured live below.  None if no summary exists."""
    if not headline:
        return None
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_trace_stats.txt"))
                   if not re.search(r"_(cfg\d|freeze)_", os.path.basename(f)))
    if not files:
        return None
    norm = lambda k: re.sub(r"\s+", "", k)
    keys = {norm(k): k for k in prepass if prepass[k]["bytes"] > 0}
    for line in open(files[-1]).read().splitlines()[2:]:
        name = norm(re.sub(r"\(anonymous namespace\)::", "", line.split("  ")[0]).replace("void", "", 1))
        for nk, k in keys.items():
            base = nk.split("<")[0]
            if name.startswith(nk + "(") or name == nk or ("<" not in nk and re.search(r"\d+" + re.escape(base) + r"(P|ILi|\b)", name)) or \
                    ("<" in nk and name.startswith(nk[:-1] + ",")) or ("<" in nk and name.startswith(nk)):      # (trace names carry trailing template parameters)
                return k
    return None


def pmc_traffic_per_launch(family):
    """HBM bytes per launch of a kernel family from the newest committed PMC pass (profiles/*pmc_traffic.json: FETCH_SIZE / WRITE_SIZE
    collected and corrected as MI355X_MICROARCH.md prescribes); None when the family is not in it."""
    import glob
    try:
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
        if not pm:
            return None
        kk = json.load(open(pm[-1]))["kernels"]
        fam = family.replace(" ", "")
        cand = [v for k_, v in kk.items() if fam.endswith(">") and k_.startswith(fam[:-1] + ",")]      # trailing template parameters
        return (kk.get(fam) or (cand[0] if cand else None) or kk.get(fam.split("<")[0]) or {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, hw, dataset, batch=2, warm=2, timed=5):
    """CPU oracle (oracle/tuber_oracle.py: stock PyTorch CPU ops, fp32, identical graph and state dict): full training steps
    (forward, criterion, backward, clip, AdamW) on a batch of the benchmark workload, all host cores: ``warm`` untimed + ``timed``
    timed steps (bounded sample: ~1 minute of CPU work)."""
    from oracle import tuber_oracle as O
    from tubelet_transformer_amd import synth
    from tubelet_transformer_amd.tuber import build_model
    cores = min(os.cpu_count() or 1, 32)     # more intra-op threads than that makes stock CPU conv3d slower, not faster
    torch.set_num_threads(cores)
    model, _, _ = build_model(cfg)
    synth.load_name_hashed(model)
    pn = [n for n, p in model.named_parameters() if p.requires_grad]
    state = {k: (v.clone().requires_grad_(True) if k in pn else v.clone()) for k, v in model.state_dict().items()}
    del model
    clips = synth.synthetic_clips(batch, 32, hw[0], hw[1], seed=1234)
    targets = synth.synthetic_targets(batch, dataset, cfg.CONFIG.DATA.NUM_CLASSES, seed=4321, hw=hw)
    params = [state[n] for n in pn]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4)
    times = []
    for i in range(warm + timed):
        t0 = time.time()
        out = O.tuber_forward(state, cfg, clips, train=True)
        ld, _ = O.set_criterion(cfg, out, targets)
        loss = O.total_loss(cfg, ld)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.1)
        opt.step()
        times.append(time.time() - t0)
    dt = sum(times[warm:]) / timed
    return {"value": round(batch / dt, 5), "unit": "clips/s", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
            "sample": "%d warm-up + %d timed full training steps (fwd+criterion+bwd+clip+AdamW), batch %d of the same 3x32x%dx%d workload, "
                      "fp32, dropout off; %.1f s/step (steps: %s)" % (warm, timed, batch, hw[0], hw[1], dt, " ".join("%.1f" % t for t in times))}


def timed_with_input_pipeline(step_fn, args, hw, dev, fence):
    """the same K steps with the input feed inside the timed region: every batch starts as decoded uint8 frames on the HOST
    (2 clips x 32 frames at 360x480, what datasets/ava_frame.py:133-152 hands to the transforms), goes through the reference's
    transform sequence as recorded geometry (resize to IMG_RESHAPE 288x384, flip, crop to the clip size, colour jitter, normalise)
    and ``ClipBatch.to(device)`` -- H2D of the uint8 frames + tuber_frames_resize + tuber_clip_prepare -- on a side stream, one batch
    ahead of the step that consumes it.  ``step_fn`` is called with the prepared NestedTensor."""
    import numpy as np
    from tubelet_transformer_amd import input_pipeline as P
    rng = np.random.default_rng(0)
    T, H0, W0 = 32, 360, 480
    RH, RW = hw[0] + 32, hw[1] + 44
    frames = [np.ascontiguousarray(rng.integers(0, 256, (T, H0, W0, 3), dtype=np.uint8)) for _ in range(args.batch)]
    pinned = [torch.from_numpy(f).pin_memory() for f in frames]      # what DataLoader(pin_memory=True) -> ClipBatch.pin_memory() hands over
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def prepare(i):
        clips = []
        for b, f in enumerate(frames):
            c = P.FrameClip(f)
            c.pinned = pinned[b]
            c.resize((RW, RH))
            if (i + b) % 2:
                c._hflip()
            c._crop(16, 22, hw[0], hw[1])
            c.jitter = (7, -13, 20)
            clips.append(c)
        with torch.cuda.stream(side):           # nothing to wait for: the batch is independent of the step in flight
            nt = P.ClipBatch(clips).to(dev)
            ev = torch.cuda.Event()
            ev.record(side)
        nt.tensors.record_stream(main)
        nt.mask.record_stream(main)
        return nt, ev

    nxt = prepare(0)
    for i in range(2):                       # warm-up (allocator, coefficient tables)
        nt, ev = nxt
        main.wait_event(ev)
        step_fn(nt)
        nxt = prepare(i + 1)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        nt, ev = nxt
        main.wait_event(ev)
        step_fn(nt)
        nxt = prepare(i + 3)
    fence()
    dt = time.perf_counter() - t0
    return {"ms_per_step": round(1e3 * dt / args.steps, 3), "seconds": dt,
            "feed": "page-locked uint8 frames %dx%dx%dx%d on the host (DataLoader pin_memory -> ClipBatch.pin_memory) -> ClipBatch.to(device) on a side stream (H2D %.1f MB + resize to %dx%d + flip/crop/jitter/"
                    "normalise/collate), one batch ahead" % (args.batch, T, H0, W0, args.batch * T * H0 * W0 * 3 / 1e6, RH, RW)}


def lib_md5():
    """checksum of the HIP library this process loaded -- recorded so that a number can be tied to the build it was measured on"""
    import hashlib
    from tubelet_transformer_amd import lib
    return hashlib.md5(open(lib.LIBPATH, "rb").read()).hexdigest()[:12]


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher (WORLD_SIZE / RANK unset): start the N ranks here, one process per GPU, the way
    the reference's ``pipelines/launch.py:20-50`` spawns its own workers -- the same command line re-executed with the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR 127.0.0.1 / a free MASTER_PORT).  Rank 0 inherits stdout, so its one JSON line is this
    job's stdout; the other ranks' stdout (RCCL banners) goes to stderr.  A rank that dies takes the job down instead of leaving the
    others waiting in a collective.  Returns the job's exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TUBER_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                print("bench.py: rank %d exited with code %d; stopping the other ranks" % (procs.index(p), code), file=sys.stderr, flush=True)
                for q in live:
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60, help="timed steps (default 60 = ~1 s of GPU time: long enough for a 5 s SMI sampler to see it)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="clips per GPU (reference: TRAIN.BATCH_SIZE 2)")
    ap.add_argument("--config", default="TubeR_CSN152_AVA21.yaml", help="BASELINE.json configs: TubeR_CSN50_AVA21.yaml (2), "
                    "TubeR_CSN152_AVA21.yaml (3, the headline), Tuber_CSN152_JHMDB.yaml (5: use --height 288 --width 384)")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=340)
    ap.add_argument("--pretrained-freeze", action="store_true", help="freeze stem + layer1 + layer2 like the pretrained recipe "
                    "(load_csn_mat, ir_CSN_152.py:251-254,301-303): the 704-GFLOP/clip workload of every real training run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="issue every launch from Python instead of replaying the captured hipGraphs")
    ap.add_argument("--no-input-pipeline", action="store_true", help="skip the second measurement with the uint8 input feed inside the timed loop (default on the headline command)")
    ap.add_argument("--with-input-pipeline", action="store_true", help="after the headline measurement, time the same steps again with every "
                    "batch fed as uint8 frames through input_pipeline.ClipBatch.to(device) (H2D + resize + flip/crop/jitter/normalise/collate "
                    "on a side stream, double-buffered) and report both figures")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        raise SystemExit(spawn_ranks(args.gpus))      # the driver's `python bench.py --gpus N`: no launcher needed

    import torch.distributed as dist
    from tubelet_transformer_amd import lib, synth
    from tubelet_transformer_amd.config import load_cfg
    from tubelet_transformer_amd.training import GraphedTrainStep, build_optimizer, deploy_model, train_step
    from tubelet_transformer_amd.tuber import build_model

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus == 1 and world == 1:
            pass
        else:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: either leave the launcher environment unset (bench.py starts its own ranks) or "
                             "launch with python -m torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if os.environ.get("TUBER_SHARE_GPU"):         # logic test of the N > 1 path on a one-GPU box (ranks share cuda:0; use gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # world > 1, or the exact process shape of one rank of a multi-GPU lease minus its peers: launcher environment
    # present (RANK / WORLD_SIZE=1 / MASTER_PORT) and TUBER_FORCE_DDP=1 -> nccl process group of one rank + the own RCCL communicator
    launched_single = world == 1 and "RANK" in os.environ and "MASTER_PORT" in os.environ and bool(os.environ.get("TUBER_FORCE_DDP"))
    if world > 1 or launched_single:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("TUBER_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    cfg = load_cfg(os.path.join(ROOT, "configuration", args.config))
    cfg.DDP_CONFIG.GPU = local
    torch.manual_seed(0)
    model, criterion, _ = build_model(cfg)
    synth.load_name_hashed(model)                 # random-init weights of the named architecture (no checkpoints offline)
    if args.pretrained_freeze:
        body = model.backbone.body
        for mod in (body.conv1, body.bn1, body.layer1, body.layer2):
            for p in mod.parameters():
                p.requires_grad = False
    model = deploy_model(model, cfg, True, device=dev)
    criterion.to(dev)
    model.train()
    criterion.train()
    optimizer = build_optimizer(model, cfg)
    hw = (args.height, args.width)
    clips = synth.synthetic_clips(args.batch, 32, hw[0], hw[1], seed=1234 + rank, device=dev)
    dataset = "ava" if cfg.CONFIG.DATA.DATASET_NAME == "ava" else "jhmdb"
    targets = synth.synthetic_targets(args.batch, dataset, cfg.CONFIG.DATA.NUM_CLASSES, seed=4321 + rank, device=dev, hw=hw)
    max_norm = cfg.CONFIG.LOSS_COFS.CLIPS_MAX_NORM

    def eager_step(samples=None):
        return train_step(model, criterion, optimizer, clips if samples is None else samples, targets, max_norm)

    mode = "eager"
    step = eager_step
    if not args.eager:
        try:
            graphed = GraphedTrainStep(model, criterion, optimizer, max_norm)
            graphed(clips, targets)
            torch.cuda.synchronize()
            mode = "hipgraph"
            # inputs resident in HBM when the timed region starts (the contract): the synthetic batch lives IN the buffers the captured
            # step reads -- what an input pre-pass writing in place does (GraphedTrainStep.input_buffers) -- so no 67 MB device-to-device
            # copy of an unchanged batch is timed.  The first call above copied it there.
            bufs = graphed.input_buffers(clips.shape)
            resident = clips if bufs is None else bufs[0]
            from tubelet_transformer_amd.misc import NestedTensor
            resident_batch = resident if bufs is None else NestedTensor(bufs[0], bufs[1])

            def step(samples=None):
                return graphed(resident_batch if samples is None else samples, targets)
        except Exception as e:      # capture is an optimisation, never a requirement
            print("hipGraph capture failed (%s: %s); running eager" % (type(e).__name__, e), file=sys.stderr, flush=True)
            torch.cuda.synchronize()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    headline_run = args.config == "TubeR_CSN152_AVA21.yaml" and hw == (256, 340) and not args.pretrained_freeze

    for _ in range(args.warmup):
        step()
    dominant, prepass = None, None
    if not args.no_roofline:
        eager_step()                                    # untimed: first eager step after the capture (allocator warm-up)
        torch.cuda.synchronize()
        timer = LaunchTimer()
        lib.set_launch_hook(timer)
        eager_step()
        torch.cuda.synchronize()
        lib.set_launch_hook(None)
        prepass = timer.summary()
        if os.environ.get("TUBER_BENCH_SHAPES") and rank == 0:
            print("\n".join(timer.by_shape(int(os.environ["TUBER_BENCH_SHAPES"]))), file=sys.stderr, flush=True)
        dominant = dominant_from_trace(prepass, headline_run) or max((k for k in prepass if prepass[k]["bytes"] > 0), key=lambda k: prepass[k]["ms"])
    fence()
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        loss, _ = step()
    fence()
    dt = time.perf_counter() - t0
    # HIP-event timing of the dominant kernel family: the same launches, same stream, same inputs, issued eagerly right after the
    # timed region (launches replayed from a hipGraph cannot be bracketed by events individually)
    timer = LaunchTimer(only=dominant) if dominant else None
    if timer:
        lib.set_launch_hook(timer)
        for _ in range(min(args.steps, 3)):
            eager_step()
        torch.cuda.synchronize()
        lib.set_launch_hook(None)
        timed_steps = min(args.steps, 3)
    red = getattr(model.engine()[0], "reducer", None)
    comm_info = None
    if red is not None:
        # the gradient exchange as it ran in the timed region + a few measured steps (HIP events around the optimizer stream's wait
        # for the transport = EXPOSED communication time), for the first multi-GPU run to be diagnosable from its one JSON line
        red.measure = True
        if mode == "hipgraph":
            graphed.part_marks = []
        for _ in range(min(args.steps, 5)):
            step()
        fence()
        comm_info = red.describe()
        red.measure = False
        if mode == "hipgraph":
            comm_info["graph_part_ms"] = graphed.part_ms()      # [graph A, A1, ..., (B2)]: HIP events between the replays
        red.check()
        parts = 0
        if mode == "hipgraph":
            parts = max((len(g.parts) for g in graphed.graphs.values()), default=0)
        comm_info["launch"] = mode + (" (graph A | " + "".join("all-reduce | graph A%d | " % (i + 1) for i in range(parts)) + "all-reduce | graph B2)"
                                      if mode == "hipgraph" and not os.environ.get("TUBER_RCCL_IN_GRAPH") else "")
        comm_info["graph_cuts"] = parts
    pipe_info = None
    fed_d2d_ms = None
    if mode == "hipgraph" and rank == 0 and world == 1:
        # the same K steps with a batch that is NOT already in the captured input buffers: + one 67 MB device-to-device copy per step (what
        # rounds 1-4 timed; the headline since round 5 is the resident form the contract prescribes)
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(clips)
        fence()
        fed_d2d_ms = 1e3 * (time.perf_counter() - t1) / args.steps
    if args.with_input_pipeline or (headline_run and world == 1 and not args.no_input_pipeline and not args.no_roofline and mode == "hipgraph"):
        pipe_info = timed_with_input_pipeline(step, args, hw, dev, fence)
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    model.engine()[0].check_coop()            # a timed-out barrier of the cooperative decoder launch would have invalidated the steps
    ms = 1e3 * dt / args.steps
    total_clips = args.batch * world * args.steps
    headline = args.config == "TubeR_CSN152_AVA21.yaml" and hw == (256, 340)
    alg_gflop = (704.0 if args.pretrained_freeze else 981.0) if headline else None      # SURVEY.md section 8d (fwd+bwd per clip)
    line = {
        "metric": "clips/sec (TubeR CSN-152 AVA2.1 training step fwd+bwd+clip+AdamW, 32x256x340 clips; whole job)" if headline else
                  "clips/sec (%s training step fwd+bwd+clip+AdamW, 32x%dx%d clips; whole job)" % (args.config.replace(".yaml", ""), hw[0], hw[1]),
        "value": round(total_clips / dt, 3), "unit": "clips/s", "per_gpu": round(total_clips / dt / world, 3),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "timed_region_s": round(dt, 4),
        "ms_per_step_resident": round(ms, 3),
        "ms_per_step_fed_device_copy": round(fed_d2d_ms, 3) if fed_d2d_ms is not None else None,
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "%s train step, %d clips/GPU of 3x32x%dx%d, dropout on, random-init name-hashed weights%s"
                               % (args.config.replace(".yaml", ""), args.batch, hw[0], hw[1],
                                  ", stem+layer1+layer2 frozen (pretrained recipe)" if args.pretrained_freeze else ""),
                   "global_batch": args.batch * world, "parallelism": "dp%d (flat-gradient RCCL all-reduce on an own communicator + stream; three issue points: behind layer4's, layer3's and the stem's backward)" % world if world > 1 else "dp1", "launch_mode": mode,
                   "lib_md5": lib_md5()},
        "final_loss": round(float(loss.detach()), 4) if loss is not None else None,
        "alg_gflop_per_clip_fwd_bwd": alg_gflop,
        "tolerance": "bf16 path vs the fp32 oracle: err <= 2x the error of a bf16-rounded execution of the oracle + 4e-3 (logits) / 1e-3 (boxes), "
                     "caps 5e-2 / 1e-2 (DESIGN.md section 4; tests/test_fullsize_gpu.py at this size)",
        "readme_implied_gflops": round(total_clips / dt * 120.0, 1),
    }
    if rank == 0 and timer is not None:
        tsum = timer.summary()
        s = tsum[dominant]
        ach = s["bytes"] / (s["ms"] * 1e-3) / 1e9
        traffic = pmc_traffic_per_launch(dominant)      # measured HBM bytes per launch of this kernel family from the committed PMC passes (profiles/)
        mfma_util, rp_avg = None, None     # from the committed PMC / kernel-trace passes of this command (profiles/), for cross-checking
        try:
            import glob
            import json as _json
            import re
            mu = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_mfma_util.json")))
            if mu:
                mk = _json.load(open(mu[-1]))["kernels"]
                dsp = dominant.replace(",", ", ")
                cand = [v for k_, v in mk.items() if dsp.endswith(">") and k_.startswith(dsp[:-1] + ",")]
                mfma_util = (mk.get(dsp) or mk.get(dominant) or (cand[0] if cand else None) or {}).get("mfma_util")
            kt = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_trace_stats.txt")) if not re.search(r"_(cfg\d|freeze)_", os.path.basename(f)))
            if kt and headline:
                for ln in open(kt[-1]).read().splitlines()[2:]:
                    dn, tn = re.sub(r"\s+", "", dominant), re.sub(r"\s+", "", ln.split("  ")[0])
                    if dn in tn or (dn.endswith(">") and dn[:-1] + "," in tn):
                        rp_avg = float(ln.split()[-4])
                        break
        except Exception:
            pass
        # headline fraction: at the rocprofv3 average launch duration of the committed kernel-trace summary of this
        # command (profiles/) when there is one -- the figure a reader can re-derive from the repository; HIP events around EAGER launches
        # (the live measurement below, kept as frac_live_hip_events) add ~3 us of event packets to a 27 us launch
        ach_rp = s["bytes"] / s["launches"] / (rp_avg * 1e-6) / 1e9 if rp_avg else None
        line["roofline"] = {"kernel": dominant, "bound": "hbm", "achieved": round(ach_rp if ach_rp else ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round((ach_rp if ach_rp else ach) / HBM_PEAK_GBS, 4),
                            "frac_source": "alg_bytes_per_launch / rocprof_avg_launch_us (committed profiles/*_kernel_trace_stats.txt of this command)" if ach_rp
                                           else "alg_bytes_per_launch / avg_launch_us (HIP events around eager launches, this run)",
                            "achieved_live_hip_events": round(ach, 1), "frac_live_hip_events": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                            "traffic_source": "committed PMC passes of this command (profiles/*pmc_traffic.json), not collected in this run",
                            "traffic_over_alg": round(traffic / (s["bytes"] / s["launches"]), 3) if traffic else None,
                            "launches_per_step": s["launches"] // max(timed_steps, 1), "launches_timed": s["launches"], "avg_launch_us": round(1e3 * s["ms"] / s["launches"], 2),
                            "alg_bytes_per_launch": int(s["bytes"] / s["launches"]),
                            "tflops": round(s["flops"] / (s["ms"] * 1e-3) / 1e12, 2),
                            "share_of_step": round(s["ms"] / timed_steps / ms, 4),
                            "mfma_util": round(mfma_util, 4) if mfma_util is not None else None,
                            "rocprof_avg_launch_us": rp_avg,
                            "frac_at_rocprof_duration": round(s["bytes"] / s["launches"] / (rp_avg * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if rp_avg else None}
        also = {}
        for k in ALSO_TIMED:
            if k != dominant and k in tsum and tsum[k]["launches"]:
                t_ = tsum[k]
                also[k] = {"bound": "mfma" if t_["flops"] / MFMA_BF16_PEAK_TFLOPS / 1e12 > t_["bytes"] / HBM_PEAK_GBS / 1e9 else "hbm",
                           "launches": t_["launches"], "avg_launch_us": round(1e3 * t_["ms"] / t_["launches"], 2),
                           "achieved_GBps": round(t_["bytes"] / (t_["ms"] * 1e-3) / 1e9, 1), "hbm_frac": round(t_["bytes"] / (t_["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "tflops": round(t_["flops"] / (t_["ms"] * 1e-3) / 1e12, 1), "mfma_frac": round(t_["flops"] / (t_["ms"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
                tr_ = pmc_traffic_per_launch(k)
                also[k]["traffic"] = tr_
                also[k]["traffic_over_alg"] = round(tr_ / (t_["bytes"] / t_["launches"]), 3) if tr_ else None
        if also:
            line["roofline_weight_gradient_gemms"] = also
        line["kernel_breakdown_ms_per_step"] = {k: round(v["ms"], 3) for k, v in sorted(prepass.items(), key=lambda kv: -kv[1]["ms"])[:12]}
        if headline and not args.pretrained_freeze:
            line["end_to_end"] = {"hbm_frac_of_alg_bytes": round(8.4e9 * total_clips / dt / (HBM_PEAK_GBS * 1e9 * world), 4),
                                  "mfma_frac_of_alg_flops": round(981e9 * total_clips / dt / (MFMA_BF16_PEAK_TFLOPS * 1e12 * world), 4)}
    if comm_info is not None:
        line["comm"] = comm_info
    if pipe_info is not None:
        pipe_info["clips_per_s"] = round(args.batch * world * args.steps / pipe_info.pop("seconds"), 3)
        line["input_pipeline"] = pipe_info
        line["ms_per_step_fed_input_pipeline"] = pipe_info["ms_per_step"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cfg, hw, dataset, batch=args.batch)
    # RCCL prints its banner through C stdio, which a pipe holds back until exit: tear the communicators down and drain every rank's
    # C buffers first, so that the JSON line is the last line this job writes to stdout.
    if red is not None and red.comm is not None:
        red.comm.close()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
