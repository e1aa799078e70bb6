"""BASELINE.json configurations at their STATED sizes (SURVEY.md section 8, VERDICT round 1 items 1/3/5).

Yardstick for the bf16 path: the fp32 CPU oracle is the truth; a bf16-ROUNDED execution of the same oracle (every conv3d / linear
on bf16-rounded operands with a bf16-rounded result, exact accumulation) is what ideal bf16 tensor-core arithmetic gives.  The HIP
path must stay within a small factor of THAT error on the same fixture -- written down per fixture in DESIGN.md section 4 -- in
addition to the absolute caps.

  config 1  TubeR_CSN50_AVA21   'decode'   1 x 3x32x224x224     eval forward (the CPU-plumbing config's geometry through the HIP path)
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

# the build's stated bf16 tolerance (DESIGN.md section 4): yardstick factor + slack against the bf16-rounded oracle, and absolute caps
CAP = {"pred_logits": 5e-2, "pred_logits_b": 5e-2, "pred_boxes": 1e-2}
K_ROUNDED, SLACK = 2.0, {"pred_logits": 4e-3, "pred_logits_b": 4e-3, "pred_boxes": 1e-3}
# the tolerance the survey stated before any bf16 execution existed (BASELINE.md section 4): met by the eval precision mode since round 6
BASELINE_TOL = {"pred_logits": 2e-2, "pred_logits_b": 2e-2, "pred_boxes": 5e-3}

FULL = {
    "cfg1_csn50_decode_224": ("TubeR_CSN50_AVA21.yaml", 1, (224, 224), "ava"),      # BASELINE config 1's geometry: 14 x 14 grid = 196 tokens (not a multiple of 16)
    "cfg2_csn50_decode": ("TubeR_CSN50_AVA21.yaml", 2, (256, 340), "ava"),
    "cfg3_csn152_avg": ("TubeR_CSN152_AVA21.yaml", 2, (256, 340), "ava"),
    "cfg4_csn152_decode": ("TubeR_CSN152_AVA22.yaml", 1, (256, 340), "ava"),
    "cfg5_csn152_jhmdb": ("Tuber_CSN152_JHMDB.yaml", 2, (288, 384), "jhmdb"),
    # round 5: config 3 on the NON-DEGENERATE fixture (synth.SPREAD_GAINS, residual_gain 0.05, structured clips): the 30 queries' actor
    # probabilities spread over ~[0.64, 0.83] with several within the bf16 noise of the 0.8 gate of PostProcessAVA -- the gate-flip
    # count of _decision_flips is a measurement on queries that CAN flip here (the bf16-rounded oracle itself flips 2 of 30)
    "cfg3_csn152_avg_spread": ("TubeR_CSN152_AVA21.yaml", 2, (256, 340), "ava"),
}


def _build(yaml_name, dev, train=False, dropout=False, residual_gain=None, spread=False):
    cfg = load_cfg(os.path.join(ROOT, "configuration", yaml_name))
    model, crit, post = build_model(cfg)
    synth.load_name_hashed(model, residual_gain=residual_gain, spread=spread)
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
    spread = case.endswith("_spread")
    cfg, model, _, state = _build(yaml_name, dev, residual_gain=0.05 if spread else None, spread=spread)
    clips = synth.structured_clips(B, 32, hw[0], hw[1], seed=1234) if spread else synth.synthetic_clips(B, 32, hw[0], hw[1], seed=1234)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.time()
    want, _ = run_oracle(cfg, state, clips, train=False)
    rnd, _ = run_oracle(cfg, state, clips, train=False, rounded=True)
    t1 = time.time()
    from tubelet_transformer_amd import ab
    with torch.no_grad():
        got = model(clips.to(dev))                      # default: the eval precision mode (fp32 residual streams, round 6)
        with ab.override("eval_bf16_decoder"):
            got_st = model(clips.to(dev))               # fp32 residual streams only: the decoder / heads on the bf16 launch chain
        with ab.override("eval_bf16_stream"):
            got_bf = model(clips.to(dev))               # the training path's rounding points (bf16-stored block / LayerNorm outputs)
    errs = output_errors(got, want, rnd)
    errs_bf = output_errors(got_bf, want, rnd)
    print("   fp32 residual streams only (TUBER_AB=eval_bf16_decoder): %s" % {k: "%.2e" % v[0] for k, v in output_errors(got_st, want, rnd).items()})
    print("%s (%s, %dx3x32x%dx%d) eval: max abs err vs fp32 oracle  hip / bf16-rounded oracle: %s   [oracle %.1f s]" % (
        case, yaml_name, B, hw[0], hw[1], {k: "%.2e / %.2e" % v for k, v in errs.items()}, t1 - t0))
    print("   the same with bf16-stored residual streams (TUBER_AB=eval_bf16_stream, the training path's rounding points): %s" % {k: "%.2e" % v[0] for k, v in errs_bf.items()})
    scale = {}
    for k, v in flat_outputs(want).items():
        scale[k.split(".")[-1]] = max(scale.get(k.split(".")[-1], 0.0), float(np.abs(v).max()))
    print("   output scales (max |fp32 oracle|): %s" % {k: "%.2f" % v for k, v in scale.items()})
    for kind in errs:
        # absolute caps were stated for O(1..3) logits; an output with a larger range (the JHMDB 2048->2 visibility head on
        # pooled features: |logit| ~ 10) gets the cap in proportion
        gain = synth.SPREAD_GAINS["class_embed_b"] if (spread and kind == "pred_logits_b") else 1.0     # the head's weight gain scales logits and error alike
        rng = max(1.0, scale[kind] / 3.0) * gain
        # (1) the default eval path -- the EVAL PRECISION MODE of round 6 (fp32 residual streams, fp32 decoder and box / actor heads) --
        # meets the tolerance BASELINE.md section 4 / SURVEY section 8c wrote down: 2e-2 on logits, 5e-3 on boxes (x the output's range)
        eh, eb = errs[kind]
        assert eh <= BASELINE_TOL[kind] * rng, "%s: %.3e exceeds the stated tolerance %.0e x %.2f" % (kind, eh, BASELINE_TOL[kind], rng)
        # (2) the training path's rounding points (bf16-stored streams, bf16 MFMA decoder): the round-3 statement -- within 2 x the bf16-rounded
        # oracle + slack and under the caps
        eh = errs_bf[kind][0]
        assert eh <= CAP[kind] * rng, (kind, eh, scale[kind])
        assert eh <= K_ROUNDED * eb + SLACK[kind], "%s: hip %.3e vs %.1f x rounded-oracle %.3e + %.0e" % (kind, eh, K_ROUNDED, eb, SLACK[kind])
    dpb = _decision_flips(case, got, want, FULL[case][3], rnd=rnd)
    if dpb is not None:          # AVA: the actor probability that gates PostProcessAVA moves by <= 5e-3 (VERDICT r05 item 4)
        assert dpb <= 5e-3, dpb
    _decision_flips(case + " [bf16 streams]", got_bf, want, FULL[case][3])


def _decision_flips(case, got, want, dataset, rnd=None):
    """What the logit error does to the DECISIONS the post-processors take (VERDICT r03 item 7: settle the tolerance with a measurement).
    AVA (PostProcessAVA, models/criterion.py:447-482): a query's 80 scores survive iff p_b = softmax(pred_logits_b)[1] > 0.8 -- count the
    queries whose gate differs between the HIP path and the fp32 oracle, at 0.8 and (the random-weight fixture keeps p_b far from 0.8)
    at every threshold of a sweep over the observed range; and the queries whose top-scoring class differs.  JHMDB (PostProcess,
    :413-445): top class of softmax(pred_logits) per tubelet query.  A flip is legitimate only where the fp32 oracle itself is
    undecided within the stated logit tolerance; any other flip fails."""
    lg_h, lg_r = got["pred_logits"].float().cpu(), want["pred_logits"].float()
    top_h, top_r = lg_h.argmax(-1), lg_r.argmax(-1)
    srt = lg_r.sort(-1, descending=True).values
    margin = (srt[..., 0] - srt[..., 1])
    flips_top = top_h != top_r
    err_lg = float((lg_h - lg_r).abs().max())
    bad_top = int((flips_top & (margin > 2 * err_lg)).sum())
    msg = "   decisions, %s: %d queries; top-class flips %d (all with fp32 top-2 margin <= 2 x logit error %.2e: %s)" % (
        case, top_r.numel(), int(flips_top.sum()), err_lg, bad_top == 0)
    assert bad_top == 0
    if dataset == "ava":
        pb_h = got["pred_logits_b"].float().cpu().softmax(-1)[..., 1]
        pb_r = want["pred_logits_b"].float().softmax(-1)[..., 1]
        dpb = float((pb_h - pb_r).abs().max())
        gate = int(((pb_h > 0.8) != (pb_r > 0.8)).sum())
        sweep = {round(float(t), 2): int(((pb_h > t) != (pb_r > t)).sum()) for t in torch.linspace(0.05, 0.95, 19)}
        worst_t = max(sweep, key=sweep.get)
        near = int(((pb_r - 0.8).abs() <= dpb).sum())
        msg += "; p_b in [%.3f, %.3f], max |dp_b| %.2e; gate flips at 0.8: %d (queries within |dp_b| of 0.8: %d); worst threshold of the sweep %.2f: %d flips" % (
            float(pb_r.min()), float(pb_r.max()), dpb, gate, near, worst_t, sweep[worst_t])
        # a gate can only flip for a query whose fp32 p_b lies within the measured p_b error of the threshold
        for t, n in sweep.items():
            assert n <= int(((pb_r - t).abs() <= dpb).sum())
        assert gate <= near
        if rnd is not None:      # the same count for the bf16-ROUNDED execution of the oracle: what ideal bf16 arithmetic does to the gate
            pb_b = rnd["pred_logits_b"].float().softmax(-1)[..., 1]
            msg += "; bf16-rounded oracle: max |dp_b| %.2e, gate flips at 0.8: %d" % (float((pb_b - pb_r).abs().max()), int(((pb_b > 0.8) != (pb_r > 0.8)).sum()))
        sc_h = lg_h.sigmoid() * ((pb_h > 0.8).float() * pb_h)[..., None]
        sc_r = lg_r.sigmoid() * ((pb_r > 0.8).float() * pb_r)[..., None]
        msg += "; max |d score| of PostProcessAVA %.2e" % float((sc_h - sc_r).abs().max())
        print(msg)
        return dpb
    print(msg)
    return None


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


# ------------------------------------------------------------------------------------------------------------------------------
# backward at REAL depth, falsifiable: (a) the well-conditioned deep fixture
# ------------------------------------------------------------------------------------------------------------------------------
RESIDUAL_GAIN = 0.05


def test_full_depth_backward_on_the_well_conditioned_fixture(dev):
    """CSN-152 at 2x3x32x256x340 through the WHOLE model (training-mode BatchNorm, dropout off, smooth surrogate loss) on the
    identity-dominated fixture (``synth.load_name_hashed(residual_gain=%.2f)``: every bn4.weight scaled): unlike the plain random
    weights -- where the bf16-rounded oracle itself decorrelates from fp32 in layer1-3 and the yardstick admits almost anything -- the
    rounded oracle stays within relerr 0.5 of fp32 on (nearly) every tensor here, so ALL 684 tensors are held, with NO waiver, to
    relerr(hip) <= 2 x relerr(rounded) + 0.05 and norm ratio in (0.5, 2), and cos(hip) >= 0.9 wherever cos(rounded) >= 0.95.  A
    backward that returned noise of the right magnitude for any tensor (relerr ~1.4) fails.  Real depth = the 1 GB deferred-reduce
    arena, 8-GEMM grouped launches spanning four bottlenecks, chained reduce entries, join fusion over 42 block boundaries.
    (Why not cos >= 0.99: a single random-weight bottleneck already costs ~7 %% of gradient accuracy under bf16 rounding -- ReLU-mask
    flips of pre-activations within one bf16 ulp of zero -- and the join ReLUs add to the residual-stream gradient in quadrature
    over 50 blocks; DESIGN.md section 4.)""" % RESIDUAL_GAIN
    B = 2 if host_mem_gb() > 160 else 1
    cfg, model, _, state = _build("TubeR_CSN152_AVA21.yaml", dev, train=True, residual_gain=RESIDUAL_GAIN)
    pn = [n for n, _ in model.named_parameters()]
    clips = synth.synthetic_clips(B, 32, 256, 340, seed=1234)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.time()
    det = lambda o: {k: (v.detach() if torch.is_tensor(v) else [{kk: vv.detach() for kk, vv in a.items()} for a in v]) for k, v in o.items()}
    o32, g32 = run_oracle(cfg, state, clips, train=True, param_names=pn, loss=surrogate)
    obf, gbf = run_oracle(cfg, state, clips, train=True, rounded=True, param_names=pn, loss=surrogate)
    o32, obf = det(o32), det(obf)
    t1 = time.time()
    store, _ = model.engine()
    store.zero_grad()
    out = model(clips.to(dev))
    surrogate(out).backward()
    torch.cuda.synchronize()
    errs = output_errors(out, o32, obf)
    print("residual_gain %.2f, batch %d, train-mode outputs hip / bf16-rounded oracle vs fp32: %s   [2 oracle fwd+bwd: %.1f s]" % (
        RESIDUAL_GAIN, B, {k: "%.2e / %.2e" % v for k, v in errs.items()}, t1 - t0))
    for kind, (eh, eb) in errs.items():
        assert eh <= 3 * eb + 2e-2, (kind, eh, eb)
    rows, _ = compare_gradients([(n, p.grad) for n, p in model.named_parameters()], g32, gbf, min_cb=None)
    report(rows, "CSN-152 %dx3x32x256x340 backward, residual gain %.2f, all tensors" % (B, RESIDUAL_GAIN))
    groups = {}
    for ch, cb, eh, eb, nr, n in rows:
        key = n.split(".")[2] if n.startswith("backbone.body.") else n.split(".")[0]
        groups.setdefault(key, []).append((eh, eb, ch, cb))
    for key, v in groups.items():
        med = lambda i: sorted(x[i] for x in v)[len(v) // 2]
        print("   %-16s tensors %3d   median relerr hip %.3f / rounded oracle %.3f   median cos hip %.4f / %.4f   min cos hip %.4f / %.4f"
              % (key, len(v), med(0), med(1), med(2), med(3), min(x[2] for x in v), min(x[3] for x in v)))
    assert len(rows) >= 620, len(rows)          # of 684; the rest has a numerically zero fp32 gradient (< 1e-5 of the global norm)
    conditioned = [r for r in rows if r[3] <= 0.5]
    print("   rounded oracle within relerr 0.5 of fp32: %d of %d tensors; cos >= 0.9: %d; cos >= 0.99: %d" % (
        len(conditioned), len(rows), sum(1 for r in rows if r[1] >= 0.9), sum(1 for r in rows if r[1] >= 0.99)))
    assert len(conditioned) >= 0.95 * len(rows), (len(conditioned), len(rows))       # the fixture does what it is for
    # the yardstick inequality for EVERY tensor; the norm ratio for every tensor the rounded oracle itself resolves (>= 95 % of
    # them by the assertion above; what is left are gradients that are numerically zero in fp32, e.g. the first decoder layer's
    # self-attention in-projection, whose queries and keys are identical for every clip)
    worse = [(n, "cos %.4f/%.4f" % (ch, cb), "relerr %.3f/%.3f" % (eh, eb), "norm %.3f" % nr) for ch, cb, eh, eb, nr, n in rows
             if eh > 2.0 * eb + 0.05 or (eb <= 0.5 and not (0.5 < nr < 2.0))]
    assert not worse, "gradients worse than 2x the bf16-rounded oracle (+0.05) or off in norm: %s" % worse[:20]
    weak = [(n, ch, cb) for ch, cb, eh, eb, nr, n in rows if cb >= 0.95 and ch < 0.9]
    assert not weak, weak[:10]
    _check_bn_buffers(cfg, model, state, clips)


# ------------------------------------------------------------------------------------------------------------------------------
# backward at REAL depth, falsifiable: (b) teacher-forced bottlenecks of the random-weight CSN-152 at the BASELINE size
# ------------------------------------------------------------------------------------------------------------------------------
def _rows(t):
    """fp32 NCDHW -> bf16 NDHWC rows [B*T*H*W, C] (the layout of the HIP path)"""
    return t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1]).to(torch.bfloat16).contiguous()


def _unrows(r, like):
    B, C, T, H, W = like.shape
    return r.float().view(B, T, H, W, C).permute(0, 4, 1, 2, 3)


def test_teacher_forced_bottleneck_gradients_at_real_depth(dev):
    """Every one of the 50 bottlenecks of CSN-152 at 2x3x32x256x340 with the random-weight fixture, conditioned independently of
    depth: the fp32 oracle's autograd supplies each block's input x_i and output gradient dy_i; the HIP path runs the block's forward
    + backward from (x_i, dy_i) through CSNRunner's ordinary code path (queued / grouped dW launches, deferred reductions, and -- for
    the multi-block segments -- the conv1-dgrad + join fusion and the 8-GEMM launches spanning four bottlenecks) and must match the
    oracle's dW / dgamma / dbeta / dx with the bf16-rounded oracle of the SAME segment as the yardstick:
    relerr(hip) <= 2 x relerr(rounded) + 0.05 (at most 0.5 % of the tensors up to 3 x), norm ratio in (0.5, 2) -- no waiver --,
    cos >= 0.9 wherever the rounded oracle has >= 0.95.  Reference: models/backbones/ir_CSN_152.py:70-90,142-170."""
    from oracle import tuber_oracle as O
    from parity_util import grad_row, rounded_convs
    B = 2 if host_mem_gb() > 100 else 1
    cfg, model, _, state = _build("TubeR_CSN152_AVA21.yaml", dev, train=True)
    store, runner = model.engine()
    P = "backbone.body"
    bstate = {k: v for k, v in state.items() if k.startswith(P + ".")}
    pn = [k for k in bstate if "running" not in k and "num_batches" not in k]
    clips = synth.synthetic_clips(B, 32, 256, 340, seed=1234)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    # --- fp32 oracle, whole body, capturing (x_i, dy_i, dx_i) of every bottleneck ------------------------------------------------
    recs = []
    orig = O.bottleneck

    def capture(st, p, x, stride, tstride, has_ds, train):
        y = orig(st, p, x, stride, tstride, has_ds, train)
        rec = {"p": p, "x": x.detach(), "args": (stride, tstride, has_ds)}
        y.register_hook(lambda g, rec=rec: rec.__setitem__("dy", g.detach().clone()))
        x.register_hook(lambda g, rec=rec: rec.__setitem__("dx", g.detach().clone()))
        recs.append(rec)
        return y
    st32 = {k: (v.clone().requires_grad_(True) if k in pn else v.clone()) for k, v in bstate.items()}
    t0 = time.time()
    O.bottleneck = capture
    try:
        feat = O.csn_body(st32, P, clips, "CSN-152", cfg.CONFIG.MODEL.LAST_STRIDE, True)
        (feat * torch.randn(feat.shape, generator=torch.Generator().manual_seed(5))).sum().backward()
    finally:
        O.bottleneck = orig
    g32 = {k: st32[k].grad for k in pn}
    del feat
    assert len(recs) == 50 and all("dy" in r and "dx" in r for r in recs)
    print("fp32 oracle body fwd+bwd with per-block capture: %.1f s" % (time.time() - t0))
    # segments: every block alone, then groups (layer1 | layer2 in two fours | layer3 in nine fours | layer4)
    segs = [(i, i + 1) for i in range(50)] + [(0, 3), (3, 7), (7, 11)] + [(11 + 4 * k, 15 + 4 * k) for k in range(9)] + [(47, 50)]
    rows, worse, weak = [], [], []
    store.refresh()
    t0 = time.time()
    for lo, hi in segs:
        names = [k for k in pn if any(k.startswith(recs[i]["p"] + ".") for i in range(lo, hi))]
        # yardstick: the bf16-rounded oracle, teacher-forced on the same segment
        stb = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in bstate.items()
               if any(k.startswith(recs[i]["p"] + ".") for i in range(lo, hi))}
        xb = recs[lo]["x"].to(torch.bfloat16).float().requires_grad_(True)
        with rounded_convs():
            yb = xb
            for i in range(lo, hi):
                yb = orig(stb, recs[i]["p"], yb, *recs[i]["args"], True)
            yb.backward(recs[hi - 1]["dy"].to(torch.bfloat16).float())
        # HIP: same inputs, CSNRunner's own forward / backward on blocks [lo, hi)
        x = recs[lo]["x"]
        store.begin_step(True)
        store.zero_grad()
        y, _, saved = runner.run_blocks(_rows(x).to(dev), (x.shape[0], x.shape[2], x.shape[3], x.shape[4]), lo, hi, train=True)
        dx = runner.backward_blocks(saved, _rows(recs[hi - 1]["dy"]).to(dev))
        torch.cuda.synchronize()
        params = dict(model.named_parameters())
        tag = "block %d" % lo if hi == lo + 1 else "blocks %d-%d" % (lo, hi - 1)
        items = [(n, params[n].grad, g32[n], stb[n].grad) for n in names] + [(recs[lo]["p"] + ".dx", _unrows(dx.cpu(), x), recs[lo]["dx"], xb.grad)]
        for n, h, a, b in items:
            ch, cb, eh, eb, nr = grad_row(h, a, b)
            rows.append((ch, cb, eh, eb, nr, tag + " " + n))
            if eh > 2.0 * eb + 0.05 or not (0.5 < nr < 2.0):
                worse.append((tag, n, "cos %.4f/%.4f" % (ch, cb), "relerr %.3f/%.3f" % (eh, eb), "norm %.3f" % nr,
                              eh > 3.0 * eb + 0.05 or not (0.5 < nr < 2.0)))
            if cb >= 0.95 and ch < 0.9:
                weak.append((tag, n, ch, cb))
    print("50 single-block + %d multi-block teacher-forced segments: %.1f s" % (len(segs) - 50, time.time() - t0))
    rows.sort()
    report(rows, "CSN-152 %dx3x32x256x340 teacher-forced bottlenecks" % B)
    for stage, (a, b) in (("layer1", (0, 3)), ("layer2", (3, 11)), ("layer3", (11, 47)), ("layer4", (47, 50))):
        sel = [r for r in rows if r[5].startswith("block ") and a <= int(r[5].split()[1]) < b]
        multi = [r for r in rows if r[5].startswith("blocks ") and a <= int(r[5].split()[1].split("-")[0]) < b]
        med = lambda v, i: sorted(x[i] for x in v)[len(v) // 2]
        print("   %s: single-block tensors %3d  median relerr hip %.4f / rounded %.4f, min cos hip %.4f | multi-block tensors %3d  median relerr %.4f / %.4f, min cos hip %.4f"
              % (stage, len(sel), med(sel, 2), med(sel, 3), min(r[0] for r in sel), len(multi), med(multi, 2), med(multi, 3), min(r[0] for r in multi)))
    single = [r for r in rows if r[5].startswith("block ")]
    multi = [r for r in rows if r[5].startswith("blocks ")]
    good = sum(1 for r in single if r[1] >= 0.99)
    print("   rounded oracle: cos >= 0.99 on %d of %d single-block tensors; min cos over the %d multi-block tensors %.4f"
          % (good, len(single), len(multi), min(r[1] for r in multi)))
    # the fixture is well-conditioned (one bottleneck deep / four deep), so the gates below bind for every tensor
    assert good >= 0.95 * len(single), (good, len(single))
    assert min(r[1] for r in multi) >= 0.9
    # The yardstick's BACKWARD is fp32 (only its forward operands are rounded) while the HIP path also stores every gradient tensor as
    # bf16, so a handful of cancellation-heavy BatchNorm scale gradients sit just outside 2x: at most 0.5 % of the tensors may, and
    # none beyond 3x (+0.05) or outside the norm window -- noise of the right magnitude (relerr ~1.4) fails either way
    print("   outside 2x (+0.05): %d of %d tensors: %s" % (len(worse), len(rows), [w[:5] for w in worse[:8]]))
    assert len(worse) <= 0.005 * len(rows), "teacher-forced gradients worse than 2x the bf16-rounded oracle (+0.05): %s" % worse[:20]
    assert not [w for w in worse if w[5]], "teacher-forced gradients worse than 3x the bf16-rounded oracle (+0.05) or off in norm: %s" % [w for w in worse if w[5]][:20]
    assert not weak, weak[:10]
