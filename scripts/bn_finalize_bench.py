"""Isolated timing of tuber_bn_finalize with and without the first-stage row reduction (tuber_stat_rows_reduce).
usage: python scripts/bn_finalize_bench.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
from tubelet_transformer_amd import lib
from gemm_bench import time_it
dev = torch.device("cuda:0")
for R, C in [(688, 128), (688, 512), (5440, 64), (5440, 256), (88, 256), (88, 1024)]:
    st0, st1 = torch.randn(R, C, device=dev), torch.rand(R, C, device=dev) + 1
    g, b, rm, rv = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    outs = [torch.empty(C, device=dev) for _ in range(4)]
    t1 = time_it(lambda: lib.call("tuber_bn_finalize", st0, st1, R, C, 1e5, g, b, rm, rv, nbt, 0.1, 1e-3, *outs))
    R2 = lib.query("tuber_stat_rows_reduced", R)
    if R2 < R:
        o0, o1 = torch.empty(R2, C, device=dev), torch.empty(R2, C, device=dev)
        def two():
            lib.call("tuber_stat_rows_reduce", st0, st1, R, C, o0, o1)
            lib.call("tuber_bn_finalize", o0, o1, R2, C, 1e5, g, b, rm, rv, nbt, 0.1, 1e-3, *outs)
        t2 = time_it(two)
    else:
        t2 = float("nan")
    print("R %5d C %5d: finalize alone %.1f us; reduce + finalize %.1f us" % (R, C, t1, t2), flush=True)
