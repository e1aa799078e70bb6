"""debug probe (TN3_DBG builds): where a 128 x 128 dW workgroup spends its life -- timestamps (s_memrealtime, 10 ns) at entry, first loads
issued, first tile parked, after 8 steps, loop end, kernel end; layer3's group of eight problems"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from tubelet_transformer_amd import lib
from tubelet_transformer_amd.engine import TnArgs
dev = torch.device("cuda:0"); BF = torch.bfloat16
probs = [(5632, 1024, 256, 1), (5632, 256, 1024, 0)] * 4
ents, keep = [], []
for M, N, K, amode in probs:
    G = torch.randn(M, N, device=dev).to(BF); A = torch.randn(M, K, device=dev).to(BF)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
    S = lib.query("tuber_gemm_tn_slabs", M, N, K)
    part = torch.empty(max(S, 1) * N * K, device=dev); out = torch.zeros(N, K, device=dev)
    ents.append(TnArgs(G.data_ptr(), N, A.data_ptr(), K, part.data_ptr(), out.data_ptr(), 2 if S > 1 else 1, M, N, K, amode, 0,
                       0, 0, 0, 0, 0, 0, 0, 0, sc.data_ptr() if amode else None, sh.data_ptr() if amode else None, None))
    keep.append((G, A, sc, sh, part, out))
arr = (TnArgs * len(ents))(*ents)
for _ in range(3):
    lib.call("tuber_gemm_tn_group", arr, len(ents))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8192 * 8))()
l = lib.load()
l.tuber_tn3_dbg_read.argtypes = [ctypes.c_void_p]
l.tuber_tn3_dbg_read(ctypes.addressof(buf))
t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
n = int((t[:, 5] > 0).sum()); t = t[:n] * 0.01          # us
t0 = t[:, 0].min()
print("workgroups %d; launch span %.1f us (first entry -> last exit)" % (n, t[:, 5].max() - t0))
print("entry time: p50 %.1f p90 %.1f max %.1f us after the first" % tuple(np.percentile(t[:, 0] - t0, [50, 90, 100])))
for name, a, b in [("issue first two steps", 0, 1), ("first tile landed + parked", 1, 2), ("8 steps", 2, 3), ("remaining steps", 3, 4), ("epilogue", 4, 5), ("whole life", 0, 5)]:
    d = t[:, b] - t[:, a]
    print("%-28s p10 %.2f p50 %.2f p90 %.2f us" % (name, *np.percentile(d, [10, 50, 90])))
