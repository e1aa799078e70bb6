"""BASELINE.json configurations at their STATED sizes.

Yardstick for the bf16 path: the fp32 CPU oracle is the truth; a bf16-ROUNDED execution of the same oracle (every conv3d / linear
on bf16-rounded operands with a bf16-rounded result, exact accumulation) is what ideal bf16 tensor-core arithmetic gives.  The HIP
path must stay within a small factor of THAT error on the same fixture -- written down per fixture in DESIGN.md section 4 -- in
addition to the absolute caps.

  config 2  TubeR_CSN50_AVA21   'decode'   2 x 3x32x256x340     eval forward + step properties
  config 3  TubeR_CSN152_AVA21  'avg'      2 x 3x32x256x340     eval forward, per-parameter backward vs oracle autograd
  config 4  TubeR_CSN152_AVA22  'decode'   1 x 3x32x256x340     eval forward
  config 5  Tuber_CSN152_JHMDB  mid-frame  2 x 3x32x288x384     eval forward + step properties (320 tubelet queries)
"""
import math
import os
import time

import numpy as np
import pytest
import torch

from parity_util import compare_gradients, flat_outputs, host_mem_gb, output_errors, report, run_oracle, surrogate
from tubelet_transformer_amd import synth
from tubelet_transformer_amd.config import load_cfg
from tubelet_transformer_amd.tuber import build_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# absolute caps (bf16 path, BASELINE.md section 4 / SURVEY.md section 8c) and the yardstick factor
CAP = {"pred_logits": 5e-2, "pred_logits_b": 5e-2, "pred_boxes": 1e-2}
K_ROUNDED, SLACK = 2.0, {"pred_logits": 4e-3, "pred_logits_b": 4e-3, "pred_boxes": 1e-3}

FULL = {
    "cfg2_csn50_decode": ("TubeR_CSN50_AVA21.yaml", 2, (256, 340), "ava"),
    "cfg3_csn152_avg": ("TubeR_CSN152_AVA21.yaml", 2, (256, 340), "ava"),
    "cfg4_csn152_decode": ("TubeR_CSN152_AVA22.yaml", 1, (256, 340), "ava"),
    "cfg5_csn152_jhmdb": ("Tuber_CSN152_JHMDB.yaml", 2, (288, 384), "jhmdb"),
}


def _build(yaml_name, dev, train=False, dropout=False):
    cfg = load_cfg(os.path.join(ROOT, "configuration", yaml_name))
    model, crit, post = build_model(cfg)
    synth.load_name_hashed(model)
    if not dropout:
        synth.zero_dropout(model)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev)
    crit.to(dev)
    model.train(train)
    crit.train(train)
    return cfg, model, crit, state


@pytest.mark.parametrize("case", list(FULL))
def test_full_size_eval_forward_vs_oracle_and_rounded_yardstick(dev, case):
    yaml_name, B, hw, _ = FULL[case]
    cfg, model, _, state = _build(yaml_name, dev)
    clips = synth.synthetic_clips(B, 32, hw[0], hw[1], seed=1234)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.time()
    want, _ = run_oracle(cfg, state, clips, train=False)
    rnd, _ = run_oracle(cfg, state, clips, train=False, rounded=True)
    t1 = time.time()
    with torch.no_grad():
        got = model(clips.to(dev))
    errs = output_errors(got, want, rnd)
    print("%s (%s, %dx3x32x%dx%d) eval: max abs err vs fp32 oracle  hip / bf16-rounded oracle: %s   [oracle %.1f s]" % (
        case, yaml_name, B, hw[0], hw[1], {k: "%.2e / %.2e" % v for k, v in errs.items()}, t1 - t0))
    scale = {}
    for k, v in flat_outputs(want).items():
        scale[k.split(".")[-1]] = max(scale.get(k.split(".")[-1], 0.0), float(np.abs(v).max()))
    print("   output scales (max |fp32 oracle|): %s" % {k: "%.2f" % v for k, v in scale.items()})
    for kind, (eh, eb) in errs.items():
        # absolute caps were stated for O(1..3) logits; an output with a larger range (the JHMDB 2048->2 visibility head on
        # pooled features: |logit| ~ 10) gets the cap in proportion
        assert eh <= CAP[kind] * max(1.0, scale[kind] / 3.0), (kind, eh, scale[kind])
        assert eh <= K_ROUNDED * eb + SLACK[kind], "%s: hip %.3e vs %.1f x rounded-oracle %.3e + %.0e" % (kind, eh, K_ROUNDED, eb, SLACK[kind])


def test_full_size_backward_per_parameter_vs_oracle(dev):
    """config 3 at BASELINE size (CSN-152, 3x32x256x340, training-mode BatchNorm, dropout off, smooth surrogate loss): EVERY
    parameter gradient of the HIP backward against fp32 autograd of the oracle, with the bf16-rounded oracle as the yardstick:
    relerr(hip) <= 2 x relerr(rounded) + 0.05, norm ratio in (0.5, 2)."""
    B = 2 if host_mem_gb() > 160 else 1            # fp32 autograd of CSN-152 at this size keeps tens of GB per clip on the host
    cfg, model, _, state = _build("TubeR_CSN152_AVA21.yaml", dev, train=True)
    pn = [n for n, _ in model.named_parameters()]
    clips = synth.synthetic_clips(B, 32, 256, 340, seed=1234)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.time()
    o32, g32 = run_oracle(cfg, state, clips, train=True, param_names=pn, loss=surrogate)
    o32 = {k: (v.detach() if torch.is_tensor(v) else [{kk: vv.detach() for kk, vv in a.items()} for a in v]) for k, v in o32.items()}
    obf, gbf = run_oracle(cfg, state, clips, train=True, rounded=True, param_names=pn, loss=surrogate)
    obf = {k: (v.detach() if torch.is_tensor(v) else [{kk: vv.detach() for kk, vv in a.items()} for a in v]) for k, v in obf.items()}
    t1 = time.time()
    store, _ = model.engine()
    store.zero_grad()
    out = model(clips.to(dev))
    surrogate(out).backward()
    torch.cuda.synchronize()
    errs = output_errors(out, o32, obf)
    print("train-mode outputs, batch %d: hip / bf16-rounded oracle vs fp32: %s   [2 oracle fwd+bwd: %.1f s]" % (
        B, {k: "%.2e / %.2e" % v for k, v in errs.items()}, t1 - t0))
    for kind, (eh, eb) in errs.items():
        assert eh <= 3 * eb + 2e-2, (kind, eh, eb)
    # ALL tensors with a non-negligible fp32 gradient (no conditioning filter): at this depth, with training-mode BatchNorm and
    # random weights, the gradient of the early stages is chaotic under bf16 rounding -- the rounded oracle itself decorrelates
    # there -- so the yardstick, not an absolute number, is the meaningful bar; per-stage medians show where that happens
    rows, worse = compare_gradients([(n, p.grad) for n, p in model.named_parameters()], g32, gbf, min_cb=None)
    report(rows, "CSN-152 %dx3x32x256x340 backward, all tensors" % B)
    groups = {}
    for ch, cb, eh, eb, nr, n in rows:
        key = n.split(".")[2] if n.startswith("backbone.body.") else n.split(".")[0]
        groups.setdefault(key, []).append((eh, eb, ch, cb))
    for key, v in groups.items():
        med = lambda i: sorted(x[i] for x in v)[len(v) // 2]
        print("   %-16s tensors %3d   median relerr hip %.3f / rounded oracle %.3f   median cos hip %.4f / %.4f" % (key, len(v), med(0), med(1), med(2), med(3)))
    well = [r for r in rows if r[3] <= 0.5]
    print("   well-conditioned tensors (rounded-oracle relerr <= 0.5): %d of %d; worst hip relerr among them %.3f" % (len(well), len(rows), max(r[2] for r in well)))
    assert len(rows) >= 600, len(rows)
    assert len(well) >= 150
    assert not worse, "gradients worse than 2x a bf16-rounded oracle (+0.05): %s" % worse[:20]
    med = len(rows) // 2
    assert sorted(r[2] for r in rows)[med] <= 1.25 * sorted(r[3] for r in rows)[med] + 0.02
    _check_bn_buffers(cfg, model, state, clips)


def _check_bn_buffers(cfg, model, state, clips):
    """BatchNorm running statistics after one training-mode forward: HIP vs the fp32 oracle, yardstick = the rounded oracle"""
    from oracle import tuber_oracle as O
    from parity_util import RoundBF
    import torch.nn.functional as F
    bufs = dict(model.named_buffers())
    st2 = {k: v.clone() for k, v in state.items()}
    st3 = {k: v.clone() for k, v in state.items()}
    with torch.no_grad():
        O.tuber_forward(st2, cfg, clips, train=True)
        oc, ol = F.conv3d, F.linear
        O.F.conv3d = lambda x, w, *a, **k: RoundBF.apply(oc(RoundBF.apply(x), RoundBF.apply(w), *a, **k))
        O.F.linear = lambda x, w, b=None: RoundBF.apply(ol(RoundBF.apply(x), RoundBF.apply(w), b))
        try:
            O.tuber_forward(st3, cfg, clips, train=True)
        finally:
            O.F.conv3d, O.F.linear = oc, ol
    worst, yard = 0.0, 0.0
    for k, v in st2.items():
        if "running_mean" in k or "running_var" in k:
            sc = max(1.0, float(v.abs().max()))
            worst = max(worst, float((bufs[k].float().cpu() - v).abs().max()) / sc)
            yard = max(yard, float((st3[k] - v).abs().max()) / sc)
        if "num_batches_tracked" in k:
            assert int(bufs[k]) == int(v), k
    print("BatchNorm running statistics: worst relative error hip %.3e / bf16-rounded oracle %.3e" % (worst, yard))
    assert worst <= 2.0 * yard + 1e-2


@pytest.mark.parametrize("yaml_name,hw,dataset", [("TubeR_CSN152_AVA21.yaml", (256, 340), "ava"), ("Tuber_CSN152_JHMDB.yaml", (288, 384), "jhmdb")])
def test_full_resolution_shallow_body_backward_per_parameter(dev, yaml_name, hw, dataset):
    """Every kernel SHAPE of the BASELINE-size step (configs 3 and 5: 2 clips of 3x32x256x340 / 3x32x288x384 -- the same per-stage
    [rows, channels] as CSN-152, whose extra depth only repeats the identity-block shapes) with a body shallow enough (CSN-TEST: 2
    bottlenecks per stage) that the gradient is well-conditioned under bf16 rounding: per parameter, relerr(hip) <= 2 x relerr(rounded
    oracle) + 0.05 and norm ratio in (0.5, 2) -- for ALL tensors, no conditioning filter -- and cos >= 0.9 wherever the rounded
    oracle reaches 0.99."""
    cfg = load_cfg(os.path.join(ROOT, "configuration", yaml_name))
    cfg.CONFIG.MODEL.BACKBONE_NAME = "CSN-TEST"
    model, _, _ = build_model(cfg)
    synth.load_name_hashed(model)
    synth.zero_dropout(model)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    pn = [n for n, _ in model.named_parameters()]
    model.to(dev).train()
    clips = synth.synthetic_clips(2, 32, hw[0], hw[1], seed=1234)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    o32, g32 = run_oracle(cfg, state, clips, train=True, param_names=pn, loss=surrogate)
    obf, gbf = run_oracle(cfg, state, clips, train=True, rounded=True, param_names=pn, loss=surrogate)
    store, _ = model.engine()
    store.zero_grad()
    out = model(clips.to(dev))
    surrogate(out).backward()
    torch.cuda.synchronize()
    det = lambda o: {k: (v.detach() if torch.is_tensor(v) else [{kk: vv.detach() for kk, vv in a.items()} for a in v]) for k, v in o.items()}
    errs = output_errors(out, det(o32), det(obf))
    print("%s CSN-TEST body at 2x3x32x%dx%d, train-mode outputs hip / rounded oracle vs fp32: %s" % (yaml_name, hw[0], hw[1], {k: "%.2e / %.2e" % v for k, v in errs.items()}))
    for kind, (eh, eb) in errs.items():
        assert eh <= 3 * eb + 2e-2, (kind, eh, eb)
    rows, worse = compare_gradients([(n, p.grad) for n, p in model.named_parameters()], g32, gbf, min_cb=None)
    report(rows, "%s shallow body, full resolution" % yaml_name)
    assert len(rows) >= 280, len(rows)
    assert not worse, "gradients worse than 2x a bf16-rounded oracle (+0.05): %s" % worse[:20]
    weak = [(n, ch, cb) for ch, cb, eh, eb, nr, n in rows if cb >= 0.99 and ch < 0.9]
    assert not weak, weak[:10]
    _check_bn_buffers(cfg, model, state, clips)


@pytest.mark.parametrize("case", ["cfg2_csn50_decode", "cfg5_csn152_jhmdb"])
def test_full_size_training_step_properties(dev, case):
    """size-independent properties of the whole fwd+bwd at the stated size, dropout on:
    (1) determinism under a fixed seed; (2) exact linearity of the backward pass in the loss scale (powers of two commute with
    rounding: fails if a gradient buffer is read before written or accumulated twice); (3) batch equivariance in eval mode;
    plus one captured optimisation step (the hipGraph path at this size) with a finite loss that moves the parameters."""
    from tubelet_transformer_amd.training import GraphedTrainStep, build_optimizer
    yaml_name, B, hw, dataset = FULL[case]
    cfg, model, crit, _ = _build(yaml_name, dev, train=True, dropout=True)
    store, _ = model.engine()
    clips = synth.synthetic_clips(B, 32, hw[0], hw[1], seed=1234, device=dev)
    targets = synth.synthetic_targets(B, dataset, cfg.CONFIG.DATA.NUM_CLASSES, seed=4321, device=dev, hw=hw)
    bn_state = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}

    def grads(scale):
        model.load_state_dict(bn_state, strict=False)
        store.manual_seed(77)
        store.zero_grad()
        out = model(clips)
        ld = crit(out, targets)
        loss = sum(ld[k] * crit.weight_dict[k] * scale for k in ld if k in crit.weight_dict)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), store.gflat.detach().clone()

    l1, g1 = grads(1.0)
    l1b, g1b = grads(1.0)
    l2, g2 = grads(2.0)
    assert math.isfinite(l1) and bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    assert l1 == l1b and torch.equal(g1, g1b), "training step is not deterministic"
    assert l2 == 2 * l1
    assert torch.equal(g2, 2 * g1), "backward is not linear in the loss scale: %d elements differ" % int((g2 != 2 * g1).sum())
    model.eval()
    with torch.no_grad():
        a = flat_outputs(model(clips))
        b = flat_outputs(model(clips.flip(0)))
    for k in a:
        assert np.array_equal(a[k], b[k][::-1] if a[k].shape[0] == 2 else b[k]), k
    model.train()
    opt = build_optimizer(model, cfg)
    step = GraphedTrainStep(model, crit, opt, cfg.CONFIG.LOSS_COFS.CLIPS_MAX_NORM)
    w0 = model.class_fc.weight.detach().clone()
    for _ in range(2):
        loss, _ = step(clips, targets)
    torch.cuda.synchronize()
    assert math.isfinite(float(loss)) and not torch.equal(w0, model.class_fc.weight.detach())
    assert bool(torch.isfinite(store.flat).all())
