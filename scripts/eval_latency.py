"""Eval forward latency of the three eval forms (round 6), BASELINE config 3 at its stated size (TubeR_CSN152_AVA21, 2 clips of 3x32x256x340),
eager launches and a captured hipGraph replay:
    default                      eval precision mode: fp32 residual streams + fp32 decoder / box / actor heads (csrc/eval_f32.hip)
    TUBER_AB=eval_bf16_decoder   fp32 residual streams only
    TUBER_AB=eval_bf16_stream    the training path's rounding points (bf16-stored streams, bf16 MFMA decoder, cooperative decoder launch)
usage: python scripts/eval_latency.py [config.yaml H W]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tubelet_transformer_amd import ab, synth                      # noqa: E402
from tubelet_transformer_amd.config import load_cfg               # noqa: E402
from tubelet_transformer_amd.tuber import build_model             # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "TubeR_CSN152_AVA21.yaml"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 340)
dev = torch.device("cuda:0")
cfg = load_cfg(os.path.join(ROOT, "configuration", name))
model, _, _ = build_model(cfg)
synth.load_name_hashed(model)
model.to(dev).eval()
clips = synth.synthetic_clips(2, 32, H, W, seed=1, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


MODES = (("eval precision mode (default)", ()), ("fp32 streams only (eval_bf16_decoder)", ("eval_bf16_decoder",)), ("bf16 streams (eval_bf16_stream)", ("eval_bf16_stream",)))
if os.environ.get("EVAL_MODES"):                                    # e.g. EVAL_MODES=0 under rocprofv3: the default mode only
    MODES = tuple(MODES[int(i)] for i in os.environ["EVAL_MODES"].split(","))
for label, sw in MODES:
    with ab.override(*sw), torch.no_grad():
        eager = timed(lambda: model(clips))
        g = torch.cuda.CUDAGraph()
        model(clips)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            model(clips)
        graph = timed(g.replay)
    print("%-44s eager %7.3f ms   hipGraph replay %7.3f ms per 2-clip batch = %.3f ms per clip, %.1f clips/s" % (label, eager, graph, graph / 2, 2e3 / graph), flush=True)
