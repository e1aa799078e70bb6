"""Phase timestamps of the cooperative decoder kernel (round 5 tuning tool, not product code): builds an INSTRUMENTED copy of
csrc/decoder_coop.hip under /tmp (s_memrealtime after every barrier, thread 0 of workgroup 0), routes the model's
tuber_decoder_coop_fwd calls to it and prints the time between consecutive barriers of one layer.   usage: python scripts/dec_probe.py"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "tubelet_transformer_amd", "csrc")
src = open(os.path.join(CSRC, "decoder_coop.hip")).read()
src = src.replace("namespace {\n\nconstexpr int E = 256", '__device__ unsigned long long dc_ts[512];\n__device__ int dc_n;\n'
                  '#define DC_STAMP() do { if (threadIdx.x == 0 && blockIdx.x == 0 && dc_n < 512) dc_ts[dc_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)\n'
                  "namespace {\n\nconstexpr int E = 256", 1)
k0 = src.index("__global__ __launch_bounds__(NT, 1) void decoder_coop_fwd_kernel")
head, body = src[:k0], src[k0:]
body = body.replace("coop_barrier(c);", "coop_barrier(c); DC_STAMP();")
body = body.replace("    const int j = blockIdx.x >> 3;\n", "    const int j = blockIdx.x >> 3;\n    if (threadIdx.x == 0 && blockIdx.x == 0) dc_n = 0;\n    DC_STAMP();\n", 1)
tail = ('\nextern "C" int dc_probe_read(unsigned long long* host, int* n) { hipMemcpyFromSymbol(n, HIP_SYMBOL(dc_n), sizeof(int)); '
        'return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dc_ts), sizeof(unsigned long long) * 512); }\n')
open("/tmp/decprobe.hip", "w").write(head + body + tail)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", "-shared",
                       "-I", CSRC, "-x", "hip", "/tmp/decprobe.hip", "-o", "/tmp/libdecprobe.so"])
P = ctypes.CDLL("/tmp/libdecprobe.so")
from tubelet_transformer_amd import lib, synth          # noqa: E402
from tubelet_transformer_amd.config import load_cfg     # noqa: E402
from tubelet_transformer_amd.tuber import build_model   # noqa: E402

lib.load()
sig = lib._sigs["tuber_decoder_coop_fwd"]
fn = P.tuber_decoder_coop_fwd
fn.restype = ctypes.c_int
fn.argtypes = [lib._ctype(t) for t, _ in sig]


def hook(name, args, launch):
    if name != "tuber_decoder_coop_fwd":
        return launch(name, *args)
    a = args + (lib.current_stream(),)
    rc = fn(*[lib._conv(v, t) for v, (t, _) in zip(a, sig)])
    assert rc == 0, rc
    return rc


dev = torch.device("cuda:0")
cfg = load_cfg(os.path.join(ROOT, "configuration", "TubeR_CSN152_AVA21.yaml"))
cfg.CONFIG.MODEL.BACKBONE_NAME = "CSN-TEST"
model, _, _ = build_model(cfg)
synth.load_name_hashed(model)
model.to(dev).train()
clips = synth.synthetic_clips(2, 32, 256, 340, seed=3, device=dev)
lib.set_launch_hook(hook)
for _ in range(3):
    out = model(clips)
torch.cuda.synchronize()
lib.set_launch_hook(None)
buf = (ctypes.c_ulonglong * 512)()
n = ctypes.c_int(0)
P.dc_probe_read(buf, ctypes.byref(n))
ts = np.array(buf[:n.value], dtype=np.int64)
d = np.diff(ts) * 0.01
names = ["start -> census barrier"] + ["A in-proj", "B self-attention", "C out-proj", "D norm1 + q-proj", "E cross-attention", "F out-proj", "G norm2 + linear1",
                                        "H linear2"] * 6 + ["I norm3 (last) + final barrier"]
print("stamps %d; total %.1f us" % (n.value, (ts[-1] - ts[0]) * 0.01))
print("%-28s %s" % ("phase (barrier to barrier)", "  ".join("layer %d" % l for l in range(6))))
print("%-28s %6.2f" % (names[0], d[0]))
for k in range(8):
    print("%-28s %s" % (names[1 + k], "  ".join("%7.2f" % d[1 + 8 * l + k] for l in range(6))))
print("%-28s %6.2f" % (names[-1], d[49] if len(d) > 49 else -1))
