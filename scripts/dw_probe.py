"""Where does a dwconv_tile_bwd_both workgroup spend its life?  (round 5 tuning tool, not product code)
Builds an INSTRUMENTED copy of csrc/dwconv_tile.hip under /tmp (s_memrealtime stamps at the phase boundaries of
dwconv_tile_bwd_both_kernel, thread 0 of every workgroup), runs it on the four stage shapes and prints the median phase durations.
The product library is not touched.   usage: python scripts/dw_probe.py"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tubelet_transformer_amd", "csrc")
src = open(os.path.join(CSRC, "dwconv_tile.hip")).read()
k0 = src.index("__global__ __launch_bounds__(512) void dwconv_tile_bwd_both_kernel")
head, body = src[:k0], src[k0:]
HALF = "--half-lds" in sys.argv
if HALF:
    # TIMING-ONLY upper bound for a bf16-parked ring (results are WRONG): every tap-loop operand read and every parking
    # store moves HALF the LDS bytes in half the instructions, the FMAs stay -- what a bf16 ring could return at most, before it pays for
    # unpacking bf16 pairs into the fp32 operands v_pk_fma_f32 needs (2 VALU per position) or for the conversion on the way in
    a0 = "                for (int i = 0; i < 10; ++i) { const float2 v = *(const float2*)(rp + i * 64); in[i] = f32x2{v.x, v.y}; }\n"
    assert body.count(a0) == 1
    body = body.replace(a0, "                for (int i = 0; i < 5; ++i) { const float2 v = *(const float2*)(rp + i * 64); in[2 * i] = f32x2{v.x, v.y}; in[2 * i + 1] = f32x2{v.y, v.x}; }\n")
    a1 = "            *(float4*)(dst + 4 * (tid + 512 * i)) = o;              // (position * 64 + quad * 4 == 4 * slot index)\n"
    assert body.count(a1) == 1
    body = body.replace(a1, "            *(float2*)(dst + 4 * (tid + 512 * i)) = make_float2(o.x + o.z, o.y + o.w);\n")
STAMPS = [  # (anchor inside the kernel, stamp id, before/after)
    ("    const TileGeom g = a.g;\n", 0, "after"),
    ("    fetch(t0 - 1, regs_a, regx_a);\n", 1, "before"),            # small loads issued
    ("    uint32_t side_nx[8];\n", 2, "before"),                      # plane fetches issued
    ("    // ---- flipped filter taps [27][64] in LDS behind the ring ----\n", 3, "before"),     # coefficients derived
    ("    for (int t = t0; t < t1; ++t) {\n", 4, "before"),          # two planes parked
    ("    // ---- workgroup reductions (the ring is dead now)", 5, "before"),                   # planes computed + stored
]
for anchor, k, where in STAMPS:
    assert body.count(anchor) == 1, anchor
    st = "    DW_STAMP(%d);\n" % k
    body = body.replace(anchor, (anchor + st) if where == "after" else (st + anchor))
end = body.index("\n}\n", body.index("(which ? a.st1 : a.st0)[(long)bx * g.C + c0 + cc] = s;"))
body = body[:end] + "\n    DW_STAMP(6);" + body[end:]
probe = ('__device__ unsigned long long dw_ts[4096 * 8];\n'
         '#define DW_STAMP(k) do { if (threadIdx.x == 0) dw_ts[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)\n')
head = head.replace("namespace {\n", probe + "namespace {\n", 1)
tail = '\nextern "C" int dw_probe_read(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dw_ts), sizeof(unsigned long long) * 4096 * 8); }\n'
open("/tmp/dwprobe.hip", "w").write(head + body + tail)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", "-shared",
                       "-I", CSRC, "-x", "hip", "/tmp/dwprobe.hip", "-o", "/tmp/libdwprobe.so"])
ctypes.CDLL(os.path.join(ROOT, "tubelet_transformer_amd", "lib", "libtuber_hip.so"), mode=ctypes.RTLD_GLOBAL)      # tuber_dw_wgrad_reduce
L = ctypes.CDLL("/tmp/libdwprobe.so")
dev = torch.device("cuda:0")
BF = torch.bfloat16
P = ctypes.c_void_p
names = ["entry->small loads issued", "plane fetches issued", "coefficients derived (3 barriers, fp64)", "taps + 2 planes parked", "plane loop (park, taps, stores)",
         "reductions + partial stores"]
for N, T, H, W, C in [(2, 32, 64, 85, 64), (2, 16, 32, 43, 128), (2, 8, 16, 22, 256), (2, 4, 16, 22, 512)]:
    M = N * T * H * W
    x, dzu, xu = (torch.randn(M, C, device=dev).to(BF) for _ in range(3))
    w = torch.randn(C, 27, device=dev) / 5
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    gamma, mean, invstd = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
    Rs = 88
    b0, b1 = torch.randn(Rs, C, device=dev), torch.randn(Rs, C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    R = L.tuber_dwconv_tile_blocks(N, T, H, W, C)
    st0, st1 = torch.empty(R, C, device=dev), torch.empty(R, C, device=dev)
    part = torch.empty(R * 27 * C, device=dev)
    p = lambda t: P(t.data_ptr())
    for _ in range(3):
        rc = L.tuber_dwconv_tile_bwd_both_bn(p(dzu), p(xu), p(b0), p(b1), Rs, ctypes.c_float(float(M)), p(gamma), p(mean), p(invstd), p(dg), p(db), p(w), p(x),
                                             p(sc), p(sh), p(out), p(st0), p(st1), p(part), N, T, H, W, C, P(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (4096 * 8))()
    L.dw_probe_read(buf)
    ts = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    nwg = min(4096, R * (C // 64))
    ts = ts[:nwg]
    d = np.diff(ts[:, :7], axis=1) * 0.01          # 100 MHz counter -> us
    life = (ts[:, 6] - ts[:, 0]) * 0.01
    span = (ts[:, 6].max() - ts[:, 0].min()) * 0.01
    print("dw bwd_both %dx%dx%dx%d C%d: %d workgroups, kernel span (first entry -> last exit) %.1f us, workgroup life median %.1f us (min %.1f max %.1f); "
          "first entry -> median entry %.1f us" % (N, T, H, W, C, nwg, span, np.median(life), life.min(), life.max(), np.median(ts[:, 0] - ts[:, 0].min()) * 0.01))
    for k, nm in enumerate(names):
        print("    %-42s median %5.2f us   (p10 %5.2f  p90 %5.2f)" % (nm, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
