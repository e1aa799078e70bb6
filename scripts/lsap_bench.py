import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scripts')
from tubelet_transformer_amd import lib
from gemm_bench import time_it
dev = torch.device("cuda:0")
for (L,B,Q,T,n) in [(6,2,15,16,3),(6,2,15,16,8),(6,2,15,16,16)]:
    cost = torch.randn(L,B,Q,T, device=dev)
    tc = torch.tensor([n]*B, dtype=torch.int32, device=dev)
    match = torch.empty(L,B,T, dtype=torch.int32, device=dev)
    print(L,B,Q,T,n, "lsap_device %.1f us" % time_it(lambda: lib.call("tuber_lsap_device", cost, tc, match, L,B,Q,T)))
