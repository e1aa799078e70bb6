"""round 6: tuber_gemm_nt on 96 x 64 tiles (plain-A shapes with >= 8 192 rows; gemm.hip: nt_use_96) against the round-5 tile choice (hook tuber_gemm_nt_96_set(0)):
outputs must be bit-identical, statistics column sums equal to fp32 summation order.  usage: python scripts/nt96_check.py"""
import torch, sys
sys.path.insert(0, ".")
from tubelet_transformer_amd import lib
dev = torch.device("cuda:0"); BF = torch.bfloat16
torch.manual_seed(0)
for (M, N, K, epi) in [(44032, 128, 512, 1), (44032, 128, 512, 2), (44032, 128, 512, 0), (44000, 128, 256, 1), (70000, 64, 128, 2)]:
    A = torch.randn(M, K, device=dev).to(BF); B = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF); Cm = torch.randn(M, N, device=dev).to(BF)
    sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    outs = []
    for on in (0, 1):
        lib.query("tuber_gemm_nt_96_set", on)
        C = torch.zeros(M, N, device=dev, dtype=BF)
        st0, st1 = torch.full((M // 32 + 8, N), 7.0, device=dev), torch.full((M // 32 + 8, N), 7.0, device=dev)
        lib.call("tuber_gemm_nt", A, K, B, K, C, N, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, epi, None, None, 0, 0, 0, st0 if epi else None, st1 if epi else None, Cm if epi == 2 else None, N, sc if epi == 2 else None, sh if epi == 2 else None, 1.0, 0.0, None, 0, None, 0, None)
        R = lib.query("tuber_gemm_nt_stat_rows", M, N)
        outs.append((C.float().clone(), st0[:R].double().sum(0).clone() if epi else None, st1[:R].double().sum(0).clone() if epi else None))
    d = float((outs[0][0] - outs[1][0]).abs().max())
    ds = max(float(((outs[0][k] - outs[1][k]).abs() / (outs[0][k].abs() + 1e-3)).max()) for k in (1, 2)) if epi else 0.0
    print(M, N, K, epi, "max |dC| %.3e, statistics rel diff %.2e" % (d, ds))
lib.query("tuber_gemm_nt_96_set", 1)
