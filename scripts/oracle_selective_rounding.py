"""Where does the bf16 path's eval error live?  (round 6; CPU only: the fp32 oracle with SELECTIVE bf16 rounding)
Runs oracle/tuber_oracle.py's forward with conv3d / linear rounding (operands and result, exact accumulation -- tests/parity_util.py's
yardstick) switched on per module group, optionally with the block outputs and / or the LayerNorm outputs additionally rounded on store
(what the training path of the HIP build does), and prints the max abs error of the three heads against the fp32 run.
usage: python scripts/oracle_selective_rounding.py [config.yaml H W] [--spread] [--groups]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
torch.set_num_threads(32)
from oracle import tuber_oracle as O
from tubelet_transformer_amd import synth
from tubelet_transformer_amd.config import load_cfg
from tubelet_transformer_amd.tuber import build_model
from parity_util import RoundBF

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "TubeR_CSN152_AVA21.yaml"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 96)
cfg = load_cfg(os.path.join(ROOT, "configuration", name))
model, _, _ = build_model(cfg)
if "--spread" in sys.argv:
    synth.load_name_hashed(model, residual_gain=0.05, spread=True)
    clips = synth.structured_clips(2, 32, H, W, seed=3)
else:
    synth.load_name_hashed(model)
    clips = synth.synthetic_clips(2, 32, H, W, seed=3)
state = {k: v.clone() for k, v in model.state_dict().items()}
oc, ol = F.conv3d, F.linear
olin, omha, obott, oln = O.linear, O.mha, O.bottleneck, O.layer_norm
G = {"on": False, "groups": set(), "body": False, "stream": False, "ln": False}

def group(p):
    if p.startswith("transformer.encoder"): return "enc"
    if p.startswith("transformer.decoder"): return "dec"
    if p.startswith("encoder.layers.0") or p.startswith("cross_attn"): return "cls"
    if p.startswith("class_embed_b") or p.startswith("bbox_embed") or p.startswith("class_fc"): return "heads"
    if p.startswith("backbone.pool"): return "pool"
    return "other:" + p

def lin(x, w, b=None):
    if G["on"]:
        return RoundBF.apply(ol(RoundBF.apply(x), RoundBF.apply(w), b))
    return ol(x, w, b)
def conv(x, w, *a, **k):
    r = (G["body"] and w.shape[1] != 256 and not (w.shape[0] == 256 and w.dim() == 5 and w.shape[1] == 2048)) 
    if w.shape[0] == 256 and w.shape[1] == 2048:      # input_proj / class_proj
        r = "proj" in G["groups"]
    if r:
        return RoundBF.apply(oc(RoundBF.apply(x), RoundBF.apply(w), *a, **k))
    return oc(x, w, *a, **k)
def linear(state_, p, x):
    G["on"] = group(p) in G["groups"]
    try: return olin(state_, p, x)
    finally: G["on"] = False
def mha(state_, p, *a, **k):
    G["on"] = group(p) in G["groups"]
    try: return omha(state_, p, *a, **k)
    finally: G["on"] = False
def bott(state_, p, x, *a, **k):
    y = obott(state_, p, x, *a, **k)
    return RoundBF.apply(y) if G["stream"] else y       # the block output STORED in bf16 (what the HIP path does)

def lnw(state_, p, x):
    y = oln(state_, p, x)
    return RoundBF.apply(y) if G["ln"] else y
def run(groups, body, stream=False, ln=False):
    G["groups"], G["body"], G["stream"], G["ln"] = set(groups), body, stream, ln
    O.F.conv3d, O.F.linear, O.linear, O.mha, O.bottleneck, O.layer_norm = conv, lin, linear, mha, bott, lnw
    try:
        with torch.no_grad():
            return O.tuber_forward({k: v.clone() for k, v in state.items()}, cfg, clips, train=False)
    finally:
        O.F.conv3d, O.F.linear, O.linear, O.mha, O.bottleneck, O.layer_norm = oc, ol, olin, omha, obott, oln

ref = run((), False)
ALL = ("enc", "dec", "cls", "heads", "pool", "proj")
rows = [("all convs / linears (the yardstick)", ALL, True, False, False), ("all + stored block outputs", ALL, True, True, False), ("all + stored LN outputs", ALL, True, False, True), ("all + both (the HIP path's rounding points)", ALL, True, True, True)]
if "--groups" in sys.argv:
    rows += [("body convs only", (), True, False, False), ("transformer + heads only", ALL, False, False, False)] + [(g + " only", (g,), False, False, False) for g in ALL] + \
            [("all but heads", ("enc", "proj", "cls", "pool", "dec"), True, False, False), ("all but decoder + heads", ("enc", "proj", "cls", "pool"), True, False, False)]
for nm, gr, body, stream, ln in rows:
    out = run(gr, body, stream, ln)
    print("%-46s" % nm, "  ".join("%s %.3e" % (k, float((out[k] - ref[k]).abs().max())) for k in ("pred_logits", "pred_boxes", "pred_logits_b")), flush=True)
