"""Round 6: can the layer3 / layer4 weight-gradient GEMM groups (1.3 ms per step that feeds nothing until the optimizer) run BESIDE the
bandwidth-bound layer2 / layer1 backward on a second stream, now that the ordering can be a device-memory counter instead of an event
(csrc/stream_flag.hip)?  Rounds 1 - 3 tried it with events / graph branches and lost 0.5 - 1.9 ms; scripts/queue_contention_probe.py
says what part of that was the event edge.  This probe times, as captured graphs on two streams with NO dependency:
    M  = a layer2 + layer1 backward-like sequence (bn_bwd_fa on the stage tensors, the one-pass depthwise backward)   [main stream]
    S  = the grouped weight-gradient launches of layer3 (16 problems per launch) x 8                                   [side stream]
alone and together.  If M || S takes about max(M, S) the overlap is real; if it takes M + S the kernels time-share the CUs.
usage: python scripts/overlap_probe.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tubelet_transformer_amd import lib
from tubelet_transformer_amd.engine import TnArgs

dev = torch.device("cuda:0")
BF = torch.bfloat16
lib.load()
keep = []


def main_seq():
    ops = []
    for M, C, R in [(44032, 512, 128), (44032, 128, 96), (348160, 256, 128), (348160, 64, 128)]:
        dz, x = torch.randn(M, C, device=dev).to(BF), torch.randn(M, C, device=dev).to(BF)
        dx = torch.empty_like(x)
        s0, s1 = torch.randn(R, C, device=dev), torch.randn(R, C, device=dev)
        gamma, mean, invstd = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        keep.append((dz, x, dx, s0, s1, gamma, mean, invstd, dg, db))
        if C >= 128:
            ops.append(lambda a=(s0, s1, R, C, float(M), gamma, mean, invstd, dg, db, dz, x, dx, M): lib.call("tuber_bn_bwd_fa", *a))
    for N, T, H, W, C in [(2, 32, 64, 85, 64), (2, 16, 32, 43, 128)]:
        M = N * T * H * W
        x, dzu, xu = (torch.randn(M, C, device=dev).to(BF) for _ in range(3))
        w = torch.randn(C, 27, device=dev) / 5
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        gamma, mean, invstd = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
        Rs = 88
        b0, b1 = torch.randn(Rs, C, device=dev), torch.randn(Rs, C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        out = torch.empty_like(x)
        R = lib.query("tuber_dwconv_tile_blocks", N, T, H, W, C)
        st0, st1 = torch.empty(R, C, device=dev), torch.empty(R, C, device=dev)
        part = torch.empty(R * 27 * C, device=dev)
        keep.append((x, dzu, xu, w, sc, sh, gamma, mean, invstd, b0, b1, dg, db, out, st0, st1, part))
        ops.append(lambda a=(dzu, xu, b0, b1, Rs, float(M), gamma, mean, invstd, dg, db, w, x, sc, sh, out, st0, st1, part, N, T, H, W, C): lib.call("tuber_dwconv_tile_bwd_both_bn", *a))
    return ops


def side_seq():
    probs = [(5632, 1024, 256, 1), (5632, 256, 1024, 0)] * 8
    ents = []
    for M, N, K, amode in probs:
        G, A = torch.randn(M, N, device=dev).to(BF), torch.randn(M, K, device=dev).to(BF)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
        S = lib.query("tuber_gemm_tn_slabs", M, N, K)
        part, out = torch.empty(max(S, 1) * N * K, device=dev), torch.zeros(N, K, device=dev)
        ents.append(TnArgs(G.data_ptr(), N, A.data_ptr(), K, part.data_ptr(), out.data_ptr(), 2 if S > 1 else 1, M, N, K, amode, 0,
                           0, 0, 0, 0, 0, 0, 0, 0, sc.data_ptr() if amode else None, sh.data_ptr() if amode else None, None))
        keep.append((G, A, sc, sh, part, out))
    arr = (TnArgs * len(ents))(*ents)
    keep.append(arr)
    return [lambda: lib.call("tuber_gemm_tn_group", arr, len(ents))]


def capture(ops, stream, reps):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        for op in ops:
            op()
        stream.synchronize()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(reps):
                for op in ops:
                    op()
    return g


sm, ss = torch.cuda.Stream(), torch.cuda.Stream()
gm = capture(main_seq(), sm, 3)          # ~3 x (bn_bwd_fa x3 + dw_both x2)
gs = capture(side_seq(), ss, 8)          # 8 grouped launches of 16 layer3 weight gradients


def timed(fn, n=20):
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def both():
    with torch.cuda.stream(sm):
        gm.replay()
    with torch.cuda.stream(ss):
        gs.replay()


def only(g, s):
    def f():
        with torch.cuda.stream(s):
            g.replay()
    return f


for rep in range(2):
    tm, ts, tb = timed(only(gm, sm)), timed(only(gs, ss)), timed(both)
    print("main (layer2 / layer1 backward-like) %.3f ms | side (8 x 16 layer3 weight-gradient GEMMs) %.3f ms | both on two streams %.3f ms   "
          "(sum %.3f, max %.3f: overlap returns %.0f %% of the shorter one)" % (tm, ts, tb, tm + ts, max(tm, ts), 100 * (tm + ts - tb) / min(tm, ts)), flush=True)
