"""phase timestamps inside decoder_fwd_kernel (a -DDEC_TIMING build of csrc/decoder.hip): cycles between the workgroup barriers of layer 0 / 1"""
import ctypes, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
so = os.path.join(tempfile.gettempdir(), "libdec_timing.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-DDEC_TIMING", "-shared", "-x", "hip",
                       os.path.join(ROOT, "scripts", "microbench", "decoder_single_wg.hip"), "-I", os.path.join(ROOT, "tubelet_transformer_amd", "csrc"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
E, Q, B, Lm, NL = 256, 15, 2, 352, 6
BF = torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.05).to(dev)
keep, ptrs = [], []
for l in range(NL):
    t = [rnd(768, E).to(BF), rnd(768), rnd(E, E).to(BF), rnd(E), torch.ones(E, device=dev), torch.zeros(E, device=dev), rnd(E, E).to(BF), rnd(E),
         rnd(E, E).to(BF), rnd(E), torch.ones(E, device=dev), torch.zeros(E, device=dev), rnd(2048, E).to(BF), rnd(2048), rnd(E, 2048).to(BF), rnd(E),
         torch.ones(E, device=dev), torch.zeros(E, device=dev), rnd(B * Lm, 512).to(BF)]
    keep.append(t)
    ptrs += [x.data_ptr() for x in t]
arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
gN, bN = torch.ones(E, device=dev), torch.zeros(E, device=dev)
qpos = rnd(Q, E).to(BF)
hs = torch.empty(NL * B * Q, E, dtype=BF, device=dev)
ts = torch.zeros(NL * B * Q * E, dtype=torch.float32, device=dev)
lib.tuber_decoder_fwd.restype = ctypes.c_int
args = [arr, NL, ctypes.c_void_p(gN.data_ptr()), ctypes.c_void_p(bN.data_ptr()), ctypes.c_void_p(qpos.data_ptr()), ctypes.c_long(512), None, B, Q, Lm,
        ctypes.c_void_p(hs.data_ptr()), ctypes.c_void_p(ts.data_ptr()), ctypes.c_float(0.0), ctypes.c_float(0.0), None, ctypes.c_ulonglong(0),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)]
for _ in range(3):
    rc = lib.tuber_decoder_fwd(*args)
torch.cuda.synchronize()
assert rc == 0, rc
t = ts.view(torch.int64)[:200].cpu().numpy()
n = int((t != 0).sum())
d = np.diff(t[:n])
names = ["in-proj", "self-attn", "out-proj1", "norm1", "operands", "q-proj", "cross-attn", "out-proj2", "norm2", "operands", "ffn1 h0", "ffn2 h0", "ffn1 h1", "ffn2 h1",
         "residual", "norm3", "hs + operands"]
per = len(names)
print("timestamps %d, cycles per phase (layer 0 | layer 1 | layer 5), s_memtime ticks:" % n)
for i, nm in enumerate(names):
    row = [d[i + per * l] if i + per * l < len(d) else -1 for l in (0, 1, 5)]
    print("  %-14s %8d %8d %8d" % (nm, *row))
print("  layer total    %8d %8d" % (d[:per].sum(), d[per:2 * per].sum()))
