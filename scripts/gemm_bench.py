"""Isolated timing of tuber_gemm_nt / tuber_gemm_tn on the backbone's shapes under forced tile configurations.
usage: python scripts/gemm_bench.py [nt|tn]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubelet_transformer_amd import lib  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
NT_SHAPES = [  # (M, N, K, amode, epi, residual)
    (5632, 256, 1024, 0, 1, 0), (5632, 1024, 256, 1, 1, 0), (5632, 256, 1024, 0, 2, 0), (5632, 1024, 256, 0, 0, 0),
    (44032, 128, 512, 0, 1, 0), (44032, 512, 128, 1, 1, 0), (44032, 128, 512, 0, 2, 0), (44032, 512, 128, 0, 0, 0),
    (348160, 64, 256, 0, 1, 0), (348160, 256, 64, 1, 1, 0), (348160, 64, 256, 0, 2, 0), (348160, 256, 64, 0, 0, 0),
    (704, 512, 2048, 0, 1, 0), (704, 2048, 512, 1, 1, 0), (704, 256, 256, 0, 0, 0), (30, 256, 256, 0, 0, 0),
    (16896, 2048, 512, 0, 0, 0),
    (5632, 256, 2048, 0, 2, 0), (5632, 1024, 512, 0, 0, 0),  # K-concatenated BN-backward variants
    (16896, 2048, 256, 0, 0, 0), (16896, 256, 2048, 0, 0, 0), (16896, 2048, 256, 0, 2, 0), (16896, 256, 256, 0, 0, 0), (16896, 768, 256, 0, 0, 0),  # class branch
    (30, 256, 2048, 0, 0, 0), (704, 256, 2048, 0, 0, 0), (30, 256, 2048, 0, 2, 0), (704, 256, 2048, 0, 2, 0), (30, 2048, 256, 0, 0, 0), (704, 2048, 256, 0, 0, 0),
]


def time_it(fn, iters=20, reps=5):
    """GPU-side time per launch: `iters` back-to-back launches captured in a hipGraph (no host launch cost), replayed."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3


def bench_nt(cfgs):
    print("%-34s" % "shape (M N K amode epi)" + "".join("  cfg%d us" % c for c in cfgs))
    for M, N, K, amode, epi, res in NT_SHAPES:
        A = torch.randn(M, K, device=dev).to(BF)
        B = torch.randn(N, K, device=dev).to(BF)
        C = torch.empty(M, N, device=dev, dtype=BF)
        Cm = torch.randn(M, N, device=dev).to(BF)
        sc, sh = torch.rand(max(K, N), device=dev) + 0.5, torch.randn(max(K, N), device=dev)
        st0, st1 = torch.empty(M // 32 + 8, N, device=dev), torch.empty(M // 32 + 8, N, device=dev)
        row = "%-34s" % ("%d %d %d %d %d" % (M, N, K, amode, epi))
        for c in cfgs:
            lib.call("tuber_gemm_nt_set_cfg", c)

            def fn():
                lib.call("tuber_gemm_nt", A, K, B, K, C, N, M, N, K, amode, sc if amode else None, sh if amode else None,
                         0, 0, 0, 0, 0, 0, 0, 0, 0, epi, None, None, 0, 0, 0, st0 if epi else None, st1 if epi else None,
                         Cm if epi == 2 else None, N, sc if epi == 2 else None, sh if epi == 2 else None, 1.0, 0.0, None, 0, None, 0, None)
            row += "  %7.1f" % time_it(fn)
        if "--blas" in sys.argv:          # the vendor library on the same shape: a yardstick for the tile experiments, never the product path
            row += "  blas %7.1f" % time_it(lambda: torch.matmul(A, B.t()))
        by = 2 * (M * K + N * K + M * N) + (2 * M * N if epi == 2 else 0)
        print(row + "   | alg %.1f MB, %.2f GF" % (by / 1e6, 2 * M * N * K / 1e9), flush=True)
    lib.call("tuber_gemm_nt_set_cfg", -1)


def bench_wsk96():
    """the layer3 / layer4 long-K 1x1x1 convs (conv1 forward with statistics, conv4 data gradient with the ReLU-BN mask) on 64-row and on
    96-row wave-split-K tiles (352 vs 236 / 240 workgroups for 256 CUs)"""
    shapes = [(5632, 256, 1024, 1), (5632, 256, 1024, 2), (5632, 256, 1024, 0), (2816, 512, 2048, 1), (2816, 512, 2048, 2)]
    for M, N, K, epi in shapes:
        A = torch.randn(M, K, device=dev).to(BF)
        B = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        Cm = torch.randn(M, N, device=dev).to(BF)
        sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        outs = []
        row = "%-24s" % ("%d %d %d epi %d" % (M, N, K, epi))
        for on in (0, 1):
            lib.query("tuber_gemm_nt_wsk96_set", on)
            C = torch.zeros(M, N, device=dev, dtype=BF)
            st0, st1 = torch.full((M // 32 + 8, N), 7.0, device=dev), torch.full((M // 32 + 8, N), 7.0, device=dev)

            def fn():
                lib.call("tuber_gemm_nt", A, K, B, K, C, N, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, epi, None, None, 0, 0, 0,
                         st0 if epi else None, st1 if epi else None, Cm if epi == 2 else None, N, sc if epi == 2 else None, sh if epi == 2 else None,
                         1.0, 0.0, None, 0, None, 0, None)
            row += "  %s %6.2f us" % ("96-row" if on else "64-row", time_it(fn))
            R = lib.query("tuber_gemm_nt_stat_rows", M, N)
            outs.append((C.float().clone(), st0[:R].double().sum(0).clone() if epi else None, st1[:R].double().sum(0).clone() if epi else None))
        d = float((outs[0][0] - outs[1][0]).abs().max())
        ds = max(float(((outs[0][k] - outs[1][k]).abs() / (outs[0][k].abs() + 1e-3)).max()) for k in (1, 2)) if epi else 0.0
        print(row + "   max |dC| %.3e, statistics rel diff %.2e" % (d, ds), flush=True)
    lib.query("tuber_gemm_nt_wsk96_set", 1)


TN_SHAPES = [(30, 256, 256), (704, 256, 256), (704, 2048, 256), (704, 256, 2048), (30, 2048, 256), (180, 256, 256),
             (5632, 1024, 256), (5632, 256, 1024), (44032, 512, 128), (44032, 128, 512), (348160, 256, 64), (348160, 64, 256),
             (16896, 768, 256), (16896, 256, 2048), (16896, 2048, 512), (2816, 2048, 512)]


def bench_tn():
    print("%-28s %8s %6s" % ("shape (M N K)", "us", "slabs"))
    for M, N, K in TN_SHAPES:
        G = torch.randn(M, N, device=dev).to(BF)
        A = torch.randn(M, K, device=dev).to(BF)
        S = lib.query("tuber_gemm_tn_slabs", M, N, K)
        part = torch.empty(S * N * K, device=dev)
        out = torch.zeros(N, K, device=dev)

        def fn():
            lib.call("tuber_gemm_tn", G, N, A, K, part, out, 1, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, None, None, None, None)
        print("%-28s %8.1f %6d   | %.1f MB %.2f GF" % ("%d %d %d" % (M, N, K), time_it(fn), S, (2 * M * (N + K) + 4 * N * K) / 1e6, 2 * M * N * K / 1e9), flush=True)


def bench_tn_group():
    """the grouped weight-gradient launch as the backward pass issues it: eight layer3 problems (conv4 / conv1 of four bottlenecks),
    three layer4 ones, the class-branch FFN pair"""
    import ctypes
    from tubelet_transformer_amd.engine import TnArgs
    groups = {"layer3 x8": [(5632, 1024, 256, 1), (5632, 256, 1024, 0)] * 4,
              "layer3 x16": [(5632, 1024, 256, 1), (5632, 256, 1024, 0)] * 8,
              "layer4 x6 ": [(2816, 2048, 512, 1), (2816, 512, 2048, 0)] * 3,
              "layer4 x6": [(2816, 2048, 512, 1), (2816, 512, 2048, 0)] * 3,
              "class-branch FFN": [(16896, 256, 2048, 0), (16896, 2048, 512, 0)],
              "encoder FFN x4": [(704, 256, 2048, 0), (704, 2048, 256, 0)] * 2}
    for name, probs in groups.items():
        ents, keep, by, fl = [], [], 0, 0
        for i, (M, N, K, amode) in enumerate(probs):
            G = torch.randn(M, N, device=dev).to(BF)
            A = torch.randn(M, K, device=dev).to(BF)
            sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
            S = lib.query("tuber_gemm_tn_slabs", M, N, K)
            part = torch.empty(max(S, 1) * N * K, device=dev)
            out = torch.zeros(N, K, device=dev)
            ents.append(TnArgs(G.data_ptr(), N, A.data_ptr(), K, part.data_ptr(), out.data_ptr(), 2 if S > 1 else 1, M, N, K, amode, 0,
                               0, 0, 0, 0, 0, 0, 0, 0, sc.data_ptr() if amode else None, sh.data_ptr() if amode else None, None))
            keep.append((G, A, sc, sh, part, out))
            by += 2 * M * (N + K) + 4 * N * K
            fl += 2 * M * N * K
        arr = (TnArgs * len(ents))(*ents)
        t = time_it(lambda: lib.call("tuber_gemm_tn_group", arr, len(ents)))
        print("%-20s %2d GEMMs  slabs %s  %7.1f us   %.1f MB alg -> %.2f TB/s (%.3f of 8 TB/s), %.0f TF/s" % (
            name, len(ents), [lib.query("tuber_gemm_tn_slabs", *q[:3]) for q in probs[:2]], t, by / 1e6, by / t / 1e6, by / t / 1e6 / 8.0, fl / t / 1e6), flush=True)


def bench_misc():
    for M, E in [(30, 256), (704, 256), (16896, 256)]:
        x = torch.randn(M, E, device=dev).to(BF)
        g, b = torch.ones(E, device=dev), torch.zeros(E, device=dev)
        y, xh, rstd = torch.empty_like(x), torch.empty_like(x), torch.empty(M, device=dev)
        nb = lib.query("tuber_layernorm_bwd_blocks", M)
        part = torch.empty(2 * nb * E, device=dev)
        dgb = torch.zeros(2 * E, device=dev)
        dx = torch.empty_like(x)
        t1 = time_it(lambda: lib.call("tuber_layernorm_fwd", x, x, g, b, y, E, xh, rstd, M, E, 1e-5, 0.1, None, 3))
        t2 = time_it(lambda: lib.call("tuber_layernorm_bwd", x, E, xh, rstd, g, dx, y, part, dgb, dgb.data_ptr() + 4 * E, 1, M, E, 0.1, None, 3))
        print("layernorm M%d E%d: fwd %.1f us, bwd (+reduce) %.1f us" % (M, E, t1, t2), flush=True)
    for M, C in [(30, 256), (704, 256), (704, 2048), (16896, 768)]:
        gq = torch.randn(M, C, device=dev).to(BF)
        out = torch.zeros(C, device=dev)
        part = torch.empty(lib.query("tuber_colsum_blocks", M) * C, device=dev)
        print("colsum M%d C%d: %.1f us" % (M, C, time_it(lambda: lib.call("tuber_colsum", gq, part, out, 1, M, C, C))), flush=True)


def bench_ffn():
    """the FFN GEMMs with their real epilogues (bias / ReLU / residual / dropout), on operands that are NOT cache-resident: each launch of a
    timed batch works on its own copy (the class-branch tensors are 8.6 - 69 MB; a single repeated copy sits in the 256 MB Infinity Cache)"""
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    for M, N, K in [(16896, 2048, 256), (16896, 256, 2048), (704, 2048, 256), (704, 256, 2048), (30, 256, 2048)]:
        ncopy = 6 if M > 10000 else 1
        A = [torch.randn(M, K, device=dev).to(BF) for _ in range(ncopy)]
        B = torch.randn(N, K, device=dev).to(BF) / K ** 0.5
        C = [torch.empty(M, N, device=dev, dtype=BF) for _ in range(ncopy)]
        R = [torch.randn(M, N, device=dev).to(BF) for _ in range(ncopy)]
        bias = torch.randn(N, device=dev)
        row = "ffn %5d x %4d x %4d:" % (M, N, K)
        for name, use_b, use_r, relu, drop in [("plain", 0, 0, 0, 0.0), ("bias", 1, 0, 0, 0.0), ("bias+relu", 1, 0, 1, 0.0), ("bias+relu+drop", 1, 0, 1, 0.1),
                                               ("bias+res", 1, 1, 0, 0.0), ("bias+res+drop", 1, 1, 0, 0.1)]:
            it = [0]

            def fn():
                i = it[0] % ncopy
                it[0] += 1
                lib.call("tuber_gemm_nt", A[i], K, B, K, C[i], N, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                         bias if use_b else None, R[i] if use_r else None, N, relu, 0, None, None, None, 0, None, None, 1.0, drop, seed, 7, None, 0, None)
            row += "  %s %.1f" % (name, time_it(fn, iters=12))
        print(row, flush=True)


def bench_bn_fa():
    """one-launch BatchNorm backward (derive the coefficients from R partial rows + apply) at the model's shapes"""
    for M, C, R in [(5632, 1024, 88), (5632, 256, 64), (1408, 2048, 22), (44032, 512, 128), (44032, 128, 96), (348160, 256, 128)]:
        dz, x = torch.randn(M, C, device=dev).to(BF), torch.randn(M, C, device=dev).to(BF)
        dx = torch.empty_like(x)
        s0, s1 = torch.randn(R, C, device=dev), torch.randn(R, C, device=dev)
        gamma, mean, invstd = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        mb = 3 * 2 * M * C / 1e6
        row = "bn_bwd_fa M%d C%d R%d:" % (M, C, R)
        for kr in (0, 4, 8, 11):           # rows per thread forced (0 = the launcher's heuristic)
            lib.query("tuber_bn_bwd_fa_rows_set", kr)
            t = time_it(lambda: lib.call("tuber_bn_bwd_fa", s0, s1, R, C, float(M), gamma, mean, invstd, dg, db, dz, x, dx, M))
            row += "  %s %.1f us" % ("heuristic (%d rows)" % lib.query("tuber_bn_bwd_fa_rows", M, C) if kr == 0 else "%d rows" % (16 * kr), t)
        lib.query("tuber_bn_bwd_fa_rows_set", 0)
        print(row + "  (3-pass alg %.1f MB)" % mb, flush=True)


def bench_dw():
    for N, T, H, W, C in [(2, 32, 64, 85, 64), (2, 16, 32, 43, 128), (2, 8, 16, 22, 256), (2, 4, 16, 22, 512)]:
        x = torch.randn(N, T, H, W, C, device=dev).to(BF)
        g = torch.randn(N, T, H, W, C, device=dev).to(BF)
        w = torch.randn(C, 27, device=dev) / 5
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        out = torch.empty_like(x)
        dwg = torch.zeros(C, 27, device=dev)
        R = lib.query("tuber_dwconv_tile_blocks", N, T, H, W, C)
        R0 = max(lib.query("tuber_dwconv_fwd_stat_rows", N, T, H, W), lib.query("tuber_dwconv_bwd_weight_blocks", N, T, H, W))
        st0, st1 = torch.empty(max(R, R0), C, device=dev), torch.empty(max(R, R0), C, device=dev)
        part = torch.empty(max(R, R0) * 27 * C, device=dev)
        t = [time_it(lambda: lib.call("tuber_dwconv_fwd", x, sc, sh, w, out, st0, st1, N, T, H, W, T, H, W, C, 1, 1)),
             time_it(lambda: lib.call("tuber_dwconv_tile_fwd", x, sc, sh, w, out, st0, st1, N, T, H, W, C)),
             time_it(lambda: lib.call("tuber_dwconv_bwd_data", g, w, x, sc, sh, out, st0, st1, N, T, H, W, T, H, W, C, 1, 1)),
             time_it(lambda: lib.call("tuber_dwconv_tile_bwd_data", g, w, x, sc, sh, out, st0, st1, N, T, H, W, C)),
             time_it(lambda: lib.call("tuber_dwconv_bwd_weight", g, x, sc, sh, part, dwg, 1, N, T, H, W, T, H, W, C, 1, 1)),
             time_it(lambda: lib.call("tuber_dwconv_tile_bwd_weight", g, x, sc, sh, part, dwg, 1, N, T, H, W, C))]
        mb = 2 * x.numel() * 2 / 1e6
        print("dwconv %dx%dx%dx%d C%d (%.0f MB in+out): fwd %.1f -> tile %.1f us | bwd data %.1f -> %.1f | bwd weight(+reduce) %.1f -> %.1f  [tile WGs %d x %d]"
              % (N, T, H, W, C, mb, *t, R, C // 64), flush=True)


def bench_dw_both():
    """depthwise backward with the BatchNorm fold: data-gradient kernel + weight-gradient kernel against the one-launch single-pass kernel"""
    for N, T, H, W, C in [(2, 32, 64, 85, 64), (2, 16, 32, 43, 128), (2, 8, 16, 22, 256), (2, 4, 16, 22, 512)]:
        M = N * T * H * W
        x = torch.randn(M, C, device=dev).to(BF)
        dzu = torch.randn(M, C, device=dev).to(BF)
        xu = torch.randn(M, C, device=dev).to(BF)
        w = torch.randn(C, 27, device=dev) / 5
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        gamma, mean, invstd = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
        Rs = 88
        b0, b1 = torch.randn(Rs, C, device=dev), torch.randn(Rs, C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        out = torch.empty_like(x)
        R = lib.query("tuber_dwconv_tile_blocks", N, T, H, W, C)
        Rw = lib.query("tuber_dwconv_tile_wgrad_blocks", N, T, H, W, C)
        st0, st1 = torch.empty(R, C, device=dev), torch.empty(R, C, device=dev)
        part = torch.empty(max(R, Rw) * 27 * C, device=dev)
        t = [time_it(lambda: lib.call("tuber_dwconv_tile_bwd_data_bn", dzu, xu, b0, b1, Rs, float(M), gamma, mean, invstd, dg, db, w, x, sc, sh, out, st0, st1, N, T, H, W, C)),
             time_it(lambda: lib.call("tuber_dwconv_tile_bwd_weight_bn", dzu, xu, b0, b1, Rs, float(M), gamma, mean, invstd, x, sc, sh, part, None, 2, N, T, H, W, C)),
             time_it(lambda: lib.call("tuber_dwconv_tile_bwd_both_bn", dzu, xu, b0, b1, Rs, float(M), gamma, mean, invstd, dg, db, w, x, sc, sh, out, st0, st1, part, N, T, H, W, C))]
        mb = 4 * 2 * M * C / 1e6
        print("dw bwd %dx%dx%dx%d C%d: data %.1f + weight %.1f = %.1f us  ->  one launch, one pass %.1f us  (4-pass alg %.1f MB -> %.2f TB/s)  [WGs %d x %d]"
              % (N, T, H, W, C, t[0], t[1], t[0] + t[1], t[2], mb, mb / t[2], R, C // 64), flush=True)


def bench_dw_scale():
    """Do the small-grid depthwise kernels (layer3: 64 workgroups on 256 CUs) get slower when more workgroups run beside them?"""
    for T, H, W, C in [(8, 16, 22, 256), (16, 32, 43, 128)]:
        for N in (2, 4, 8, 16):
            x = torch.randn(N, T, H, W, C, device=dev).to(BF)
            g = torch.randn(N, T, H, W, C, device=dev).to(BF)
            w = torch.randn(C, 27, device=dev) / 5
            sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
            out = torch.empty_like(x)
            dwg = torch.zeros(C, 27, device=dev)
            R = lib.query("tuber_dwconv_tile_blocks", N, T, H, W, C)
            st0, st1 = torch.empty(R, C, device=dev), torch.empty(R, C, device=dev)
            part = torch.empty(R * 27 * C, device=dev)
            t = [time_it(lambda: lib.call("tuber_dwconv_tile_fwd", x, sc, sh, w, out, st0, st1, N, T, H, W, C)),
                 time_it(lambda: lib.call("tuber_dwconv_tile_bwd_data", g, w, x, sc, sh, out, st0, st1, N, T, H, W, C)),
                 time_it(lambda: lib.call("tuber_dwconv_tile_bwd_weight", g, x, sc, sh, part, dwg, 1, N, T, H, W, C))]
            print("dw scale %dx%dx%dx%d C%d: fwd %.1f  bwd data %.1f  bwd weight(+reduce) %.1f us  [WGs %d x %d]" % (N, T, H, W, C, *t, R, C // 64), flush=True)


def bench_attn():
    import numpy as np
    H, E = 8, 256
    for B, Lq, Lk in [(48, 352, 352), (2, 352, 352), (2, 15, 352), (12, 15, 1408)]:
        q = torch.randn(Lq, B, E, device=dev).to(BF)
        k = torch.randn(Lk, B, E, device=dev).to(BF)
        v = torch.randn(Lk, B, E, device=dev).to(BF)
        o, do = torch.empty_like(q), torch.randn(Lq, B, E, device=dev).to(BF)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        lse, delta = torch.empty(B, H, Lq, device=dev), torch.empty(B, H, Lq, device=dev)
        mp = np.array([E, B, 1, 0, 1], dtype=np.int64)
        m = mp.ctypes.data
        seed = torch.full((1,), 5, dtype=torch.int64, device=dev)
        tf = time_it(lambda: lib.call("tuber_attn_fwd", q, m, k, m, v, m, o, m, lse, None, B, H, Lq, Lk, 32 ** -0.5, 0.1, seed, 3))
        tb = time_it(lambda: lib.call("tuber_attn_bwd", q, m, k, m, v, m, o, m, lse, None, do, m, dq, m, dk, m, dv, m, delta, B, H, Lq, Lk,
                                      32 ** -0.5, 0.1, seed, 3))
        fl = 4.0 * B * H * Lq * Lk * 32
        print("attention B%d Lq%d Lk%d: fwd %.1f us (%.1f TF/s), bwd %.1f us (%.1f TF/s)" % (B, Lq, Lk, tf, fl / tf / 1e6, tb, 2.5 * fl / tb / 1e6), flush=True)


def bench_stem():
    B, T, H, W = 2, 32, 256, 340
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    M, Mp = B * T * Ho * Wo, B * T * Hp * Wp
    clip = torch.randn(B, 3, T, H, W, device=dev)
    wp = torch.zeros(64, 512, device=dev, dtype=BF)
    lib.call("tuber_stem_pack_weight", torch.randn(64, 441, device=dev) / 21, wp)
    R = lib.query("tuber_stem_conv_blocks", B, T, H, W)
    c0 = torch.empty(M, 64, device=dev, dtype=BF)
    st0, st1 = torch.empty(4096, 64, device=dev), torch.empty(4096, 64, device=dev)
    g = torch.randn(M, 64, device=dev).to(BF)
    part = torch.empty(lib.query("tuber_stem_conv_wgrad_blocks", B, T, H, W) * 512 * 64, device=dev)
    dw = torch.zeros(64, 441, device=dev)
    sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    out = torch.empty(Mp, 64, device=dev, dtype=BF)
    arg = torch.empty(Mp, 64, device=dev, dtype=torch.uint8)
    gp = torch.randn(Mp, 64, device=dev).to(BF)
    dz = torch.empty(M, 64, device=dev, dtype=BF)
    t = [time_it(lambda: lib.call("tuber_stem_conv_fwd", clip, wp, c0, st0, st1, B, T, H, W), iters=5),
         time_it(lambda: lib.call("tuber_stem_conv_bwd_weight", clip, g, part, dw, 0, B, T, H, W), iters=5),
         time_it(lambda: lib.call("tuber_stem_pool_fwd", c0, sc, sh, out, arg, B * T, Ho, Wo, Hp, Wp), iters=5),
         time_it(lambda: lib.call("tuber_stem_pool_bwd", gp, arg, c0, sc, sh, dz, st0, st1, B * T, Ho, Wo, Hp, Wp), iters=5)]
    alg = [4 * clip.numel() + 2 * M * 64, 4 * clip.numel() + 2 * M * 64, 2 * M * 64 + 3 * Mp * 64, 4 * M * 64 + 3 * Mp * 64]
    for n, us, by in zip(["conv fwd", "conv wgrad (+reduce)", "pool fwd", "pool bwd"], t, alg):
        print("stem %-22s %8.1f us   alg %.0f MB -> %.0f GB/s" % (n, us, by / 1e6, by / us / 1e3), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wsk96":
        bench_wsk96()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stem":
        bench_stem()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attn":
        bench_attn()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dwscale":
        bench_dw_scale()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ffn":
        bench_ffn()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bnfa":
        bench_bn_fa()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dwboth":
        bench_dw_both()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dw":
        bench_dw()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tn":
        bench_tn()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tngroup":
        bench_tn_group()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "misc":
        bench_misc()
        sys.exit(0)
    bench_nt([int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,3,4,5".split(","))])
