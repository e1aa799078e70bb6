"""debug: per-BN batch statistics of the HIP train-mode forward vs the fp32 oracle (shallow body)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.nn.functional as F
from oracle import tuber_oracle as O
from tubelet_transformer_amd import synth
from tubelet_transformer_amd.config import load_cfg
from tubelet_transformer_amd.tuber import build_model

name = sys.argv[1] if len(sys.argv) > 1 else "CSN-TEST"
cfg = load_cfg(os.path.join(ROOT, "configuration", "TubeR_CSN152_AVA21.yaml"))
cfg.CONFIG.MODEL.BACKBONE_NAME = name
model, _, _ = build_model(cfg)
synth.load_name_hashed(model)
synth.zero_dropout(model)
state0 = {k: v.clone() for k, v in model.state_dict().items()}
clips = synth.synthetic_clips(2, 32, 64, 96, seed=99)


class RoundBF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def run(rounded):
    st = {k: v.clone() for k, v in state0.items()}
    oc, ol = F.conv3d, F.linear
    if rounded:
        O.F.conv3d = lambda x, w, *a, **k: RoundBF.apply(oc(RoundBF.apply(x), RoundBF.apply(w), *a, **k))
        O.F.linear = lambda x, w, b=None: RoundBF.apply(ol(RoundBF.apply(x), RoundBF.apply(w), b))
    try:
        with torch.no_grad():
            out = O.tuber_forward(st, cfg, clips, train=True)
    finally:
        O.F.conv3d, O.F.linear = oc, ol
    return st, out


s32, o32 = run(False)
sbf, obf = run(True)
model.cuda().train()
with torch.no_grad():
    out = model(clips.cuda())
sh = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
print("%-52s %12s %12s   (batch-stat error implied by the running buffers, x10)" % ("buffer", "hip-fp32", "bf16orc-fp32"))
for k in s32:
    if k.endswith("running_mean") or k.endswith("running_var"):
        sc = float(s32[k].abs().max()) + 1e-6
        e1 = float((sh[k] - s32[k]).abs().max()) * 10
        e2 = float((sbf[k] - s32[k]).abs().max()) * 10
        flag = "  <<<" if e1 > 5 * e2 + 1e-3 * sc else ""
        print("%-52s %12.4e %12.4e  scale %.3e%s" % (k, e1, e2, sc, flag))
for k in ("pred_logits", "pred_boxes", "pred_logits_b"):
    print(k, float((out[k].float().cpu() - o32[k]).abs().max()), float((obf[k] - o32[k]).abs().max()))
