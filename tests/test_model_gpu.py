"""End-to-end parity of the HIP model against the committed golden vectors (generated from the imported reference)
and against the CPU oracle run on the GPU box with the same name-hashed weights and seeded inputs.

Stated tolerance of the bf16 path (this build's own statement -- DESIGN.md section 4, README.md, the bench line's ``tolerance``
field; it REPLACES the 2e-2 / 5e-3 figures BASELINE.md section 4 / SURVEY.md section 8c guessed before any bf16 run existed, which an
ideally-accumulated bf16 execution of the reference graph does not meet either: 2.0e-2 .. 3.9e-2 on ``pred_logits_b``):
  eval outputs:  err <= 2x the error of a bf16-ROUNDED execution of the fp32 oracle on the same fixture + slack (4e-3 logits /
                 1e-3 boxes), AND absolute caps 5e-2 on logits (O(1..3)) / 1e-2 on boxes: the numbers per fixture are in
                 DESIGN.md section 4; the BASELINE-size cases live in tests/test_fullsize_gpu.py
  train step (golden, deep bodies): the Hungarian assignment is discontinuous and training-mode BatchNorm over the
                 few samples of the tiny fixtures amplifies bf16 rounding (a bf16-ROUNDED run of the fp32 oracle itself
                 flips 12/12 assignments and decorrelates early-layer gradients: see DESIGN.md "parity"), so the golden
                 check pins aggregate quantities: total loss within 2%, every loss term within 20%, global grad-norm
                 within 30%, per-tensor grad-norm ratios in [0.5, 2], head gradients (closest to the loss) cos >= 0.95.
  backward (shallow body, smooth loss): per-parameter relative gradient error against autograd of the fp32 oracle must
                 be <= 2x the error of a bf16-ROUNDED execution of the oracle (+0.05) -- see test_backward_vs_oracle.
"""
import math
import os

import numpy as np
import pytest
import torch

from tubelet_transformer_amd import synth
from tubelet_transformer_amd.config import load_cfg
from tubelet_transformer_amd.tuber import build_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EVAL_CASES = {
    "csn152_ava21_avg_eval_ragged": ("TubeR_CSN152_AVA21.yaml", [(64, 96), (48, 80)]),
    "csn50_ava21_decode_eval": ("TubeR_CSN50_AVA21.yaml", [(64, 96)]),
    "csn152_ava22_decode_eval": ("TubeR_CSN152_AVA22.yaml", [(64, 64)]),
    "csn152_jhmdb_eval": ("Tuber_CSN152_JHMDB.yaml", [(64, 64)]),
    "csn152_ava21_avg_eval_spread": ("TubeR_CSN152_AVA21.yaml", [(64, 96), (64, 96)]),      # synth.SPREAD_GAINS + structured clips
}
TRAIN_CASES = {
    "csn152_ava21_avg_train": ("TubeR_CSN152_AVA21.yaml", [(64, 96), (64, 96)]),
    "csn50_ava21_decode_train": ("TubeR_CSN50_AVA21.yaml", [(64, 64), (64, 64)]),
    "csn152_jhmdb_train": ("Tuber_CSN152_JHMDB.yaml", [(64, 64), (64, 64)]),
}


SPREAD_TRAIN_CASES = {
    "csn152_ava21_avg_train_spread": ("TubeR_CSN152_AVA21.yaml", [(64, 96), (64, 96)]),
    "csn50_ava21_decode_train_spread": ("TubeR_CSN50_AVA21.yaml", [(64, 64), (64, 64)]),
    "csn152_jhmdb_train_spread": ("Tuber_CSN152_JHMDB.yaml", [(64, 64), (64, 64)]),
}
SPREAD_RESIDUAL_GAIN = 0.05          # oracle/gen_golden.py: SPREAD_RESIDUAL_GAIN


def make_clips(sizes, seed):
    if len(set(sizes)) == 1:
        return synth.synthetic_clips(len(sizes), 32, sizes[0][0], sizes[0][1], seed=seed)
    return synth.synthetic_clips(len(sizes), 32, 0, 0, seed=seed, sizes=sizes)


def flat_outputs(out):
    d = {k: v.detach().float().cpu().numpy() for k, v in out.items() if k not in ("aux_outputs", "_stacked")}
    for i, a in enumerate(out.get("aux_outputs", [])):
        for k, v in a.items():
            d["aux%d.%s" % (i, k)] = v.detach().float().cpu().numpy()
    return d


def build(yaml_name, dev, train=False, spread=False):
    cfg = load_cfg(os.path.join(ROOT, "configuration", yaml_name))
    model, crit, post = build_model(cfg)
    if spread:
        synth.load_name_hashed(model, residual_gain=SPREAD_RESIDUAL_GAIN, spread=True)
    else:
        synth.load_name_hashed(model)
    synth.zero_dropout(model)            # deterministic train mode: every dropout probability -> 0
    model.to(dev)
    crit.to(dev)
    model.train(train)
    crit.train(train)
    return cfg, model, crit, post


@pytest.mark.parametrize("name", list(EVAL_CASES))
def test_eval_forward_matches_reference_golden(dev, golden_dir, name):
    yaml_name, sizes = EVAL_CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    spread = name.endswith("_spread")
    cfg, model, _, post = build(yaml_name, dev, spread=spread)
    clips = synth.structured_clips(len(sizes), 32, sizes[0][0], sizes[0][1], seed=1234) if spread else make_clips(sizes, seed=1234)
    clips = [c.to(dev) for c in clips] if isinstance(clips, list) else clips.to(dev)
    with torch.no_grad():
        out = model(clips)
    got = flat_outputs(out)
    worst = {}
    for k, v in got.items():
        err = float(np.abs(v - gold[k]).max())
        kind = k.split(".")[-1]
        worst[kind] = max(worst.get(kind, 0.0), err)
        assert np.isfinite(v).all(), k
    # yardstick: the same fixture through a bf16-ROUNDED execution of the fp32 oracle (parity_util.run_oracle), against the same golden
    from parity_util import run_oracle
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cl = [c.cpu() for c in clips] if isinstance(clips, list) else clips.cpu()
    rnd = flat_outputs(run_oracle(cfg, state, cl, train=False, rounded=True)[0])
    yard = {}
    for k, v in rnd.items():
        kind = k.split(".")[-1]
        yard[kind] = max(yard.get(kind, 0.0), float(np.abs(v - gold[k]).max()))
    print("%-32s max abs err vs reference: hip %s | bf16-rounded oracle %s" % (name, {k: "%.2e" % v for k, v in worst.items()},
                                                                              {k: "%.2e" % v for k, v in yard.items()}))
    gain_b = synth.SPREAD_GAINS["class_embed_b"] if spread else 1.0      # the actor head's weight gain scales its logits AND their error
    assert worst["pred_boxes"] <= 1e-2
    assert worst["pred_logits"] <= 5e-2
    assert worst["pred_logits_b"] <= 5e-2 * gain_b
    for kind in worst:
        assert worst[kind] <= 2.0 * yard[kind] + (1e-3 if kind == "pred_boxes" else 4e-3), (kind, worst[kind], yard[kind])
    # post-processing on the HIP outputs vs the reference's post-processing of its own outputs
    tsz = torch.as_tensor(gold["post.target_sizes"])
    scores, boxes, out_b = post["bbox"](out, tsz)
    if cfg.CONFIG.DATA.DATASET_NAME != "ava":   # softmax scores are smooth; AVA scores gate on p_b > 0.8 (discontinuous)
        assert np.abs(scores - gold["post.scores"]).max() <= 2e-2
    assert np.abs(boxes - gold["post.boxes"]).max() <= 1e-2 * float(tsz.max())
    assert np.abs(out_b - gold["post.out_b"]).max() <= 2e-2


@pytest.mark.parametrize("name", list(TRAIN_CASES))
def test_train_step_matches_reference_golden(dev, golden_dir, name):
    yaml_name, sizes = TRAIN_CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg, model, crit, _ = build(yaml_name, dev, train=True)
    ava = cfg.CONFIG.DATA.DATASET_NAME == "ava"
    clips = make_clips(sizes, seed=99).to(dev)
    targets = synth.synthetic_targets(len(sizes), "ava" if ava else "jhmdb", cfg.CONFIG.DATA.NUM_CLASSES, seed=7, hw=sizes[0],
                                      boxes_per_clip=[2, 3] if ava else None, device=dev)
    store, _ = model.engine()
    store.zero_grad()
    out = model(clips)
    got = flat_outputs(out)
    for k in ("pred_logits", "pred_boxes", "pred_logits_b"):
        print("train-mode output %-14s max abs err %.3e" % (k, float(np.abs(got[k] - gold["out." + k]).max())))
    from parity_util import criterion_probe, check_criterion_on_model_outputs
    criterion_probe(out)
    ld = crit(out, targets)
    wd = crit.weight_dict
    loss = sum(ld[k] * wd[k] for k in ld if k in wd)
    loss.backward()
    torch.cuda.synchronize()
    # the matcher -> gather -> loss chain on THESE outputs: assignment bit-exact, losses 1e-4, output gradients 2e-5 (no bf16 excuse)
    check_criterion_on_model_outputs(cfg, crit, out, targets, ld, tag=name)
    # matcher indices: identical to the reference's unless a bf16-level cost perturbation flips a near-tie
    same = 0
    total = 0
    for li, per in enumerate(crit.last_indices):
        for b, (i, j) in enumerate(per):
            total += 1
            same += int(np.array_equal(i.numpy(), gold["match.%d.%d.src" % (li, b)]) and np.array_equal(j.numpy(), gold["match.%d.%d.tgt" % (li, b)]))
    print("matcher assignments identical to the reference: %d / %d" % (same, total))
    bad = []
    for k in sorted(ld):
        if k == "class_error":
            continue
        g, r = float(ld[k]), float(gold["loss." + k])
        print("  %-14s hip %.5f  ref %.5f" % (k, g, r))
        if abs(g - r) > 0.20 * abs(r) + 1e-3:
            bad.append(k)
    print("total loss hip %.5f ref %.5f" % (float(loss), float(gold["total_loss"])))
    assert math.isfinite(float(loss))
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
    gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    print("global grad norm hip %.4f ref %.4f" % (gn, float(gold["grad_norm"])))
    names, norms = list(gold["grad_names"]), gold["grad_norms"]
    ratios = []
    for n, rn in zip(names, norms):
        hn = float(grads[str(n)].norm())
        if rn > 1e-3 * float(gold["grad_norm"]):
            ratios.append((hn / rn, str(n)))
    ratios.sort()
    print("per-parameter grad-norm ratio hip/ref over %d significant tensors: min %.3f (%s)  median %.3f  max %.3f (%s)" % (
        len(ratios), ratios[0][0], ratios[0][1], ratios[len(ratios) // 2][0], ratios[-1][0], ratios[-1][1]))
    cos_bad = []
    for k in gold.files:
        if k.startswith("grad."):
            n = k[5:]
            a, b = grads[n].flatten().double(), torch.as_tensor(gold[k]).flatten().double()
            cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
            print("  cos(grad %-44s) = %.4f   |hip| %.3e |ref| %.3e" % (n, cos, float(a.norm()), float(b.norm())))
            if n in ("class_embed_b.weight", "class_fc.bias") and cos < 0.95:
                cos_bad.append(n)
    for k in gold.files:
        if k.startswith("buf."):
            v = dict(model.named_buffers())[k[4:]].float().cpu().numpy()
            err = float(np.abs(v - gold[k]).max())
            print("  buffer %-52s max abs err %.3e" % (k[4:], err))
            assert err <= 0.15 * max(1.0, float(np.abs(gold[k]).max()))
    assert not bad, "loss terms off: %s" % bad
    assert abs(float(loss) - float(gold["total_loss"])) <= 0.02 * abs(float(gold["total_loss"]))
    assert abs(gn - float(gold["grad_norm"])) <= 0.30 * float(gold["grad_norm"])
    assert not cos_bad, "gradient direction off: %s" % cos_bad
    assert 0.5 < ratios[0][0] and ratios[-1][0] < 2.0


def _nsr(c):
    """noise-to-signal ratio of a gradient whose cosine against the truth is c (noise orthogonal to the signal)"""
    return math.sqrt(max(1.0 / max(c, 1e-3) ** 2 - 1.0, 0.0))


@pytest.mark.parametrize("name", list(SPREAD_TRAIN_CASES))
def test_train_step_on_the_spread_fixture_reproduces_the_reference_assignment(dev, golden_dir, name):
    """End-to-end training step on the NON-DEGENERATE fixture: synth.SPREAD_GAINS keep the tubelet queries apart
    (decoder states differ by ~30 % of their norm, boxes by >= 0.05), residual_gain 0.05 keeps the deep body well-conditioned, the clip
    and target seeds were chosen by oracle/gen_golden.py: spread_search so that the reference's Hungarian assignment survives a
    bf16-ROUNDED execution of the fp32 oracle on every (decoder layer, clip) problem, with every alternative's cost gap >= 2.3 x the
    perturbation that execution realises (stored per problem as ``ratio`` / ``noise``).  What is asserted:

    * the fused criterion on the model's own outputs == the oracle's criterion on the same values: assignment bit-exact, losses 1e-4,
      output gradients 2e-5 (check_criterion_on_model_outputs);
    * the assignment against the REFERENCE's, per problem: identical wherever the golden's decidability ratio is >= 6 and on at least
      9 of the 12 problems; a flip elsewhere is accepted only if the cost perturbation of the HIP run on that problem is <= 3 x the
      rounded oracle's (+ 0.02) -- i.e. explained by bf16 noise of the size ideal bf16 arithmetic has on the same fixture;
    * on the layers whose assignment is the reference's: every loss term within max(2 %, 4 x the rounded oracle's error on that term's
      family + 1 %); with all assignments identical also, each against what the FULLY rounded oracle (parity_util.run_oracle
      rounded="full": activation gradients, attention probabilities and score gradients through bf16 too) realises on this fixture:
      total loss within max(0.5 %, 2 x + 0.1 %), global gradient norm within max(5 %, 2 x + 1 %), per-tensor gradient-norm ratios
      inside its range widened by (0.67, 1.5), and for every stored gradient tensor (the stem's conv1 through 50 bottlenecks, the
      query embedding through 18 attention blocks, ...) a noise-to-signal ratio sqrt(1 / cos^2 - 1) <= 3 x the yardstick's + 0.3
      (the HIP path also keeps the residual streams in bf16, which the yardstick does not), the heads' cosines >= 0.98.
    """
    from parity_util import criterion_probe, check_criterion_on_model_outputs, matcher_problems, assignment_margin
    yaml_name, sizes = SPREAD_TRAIN_CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg, model, crit, _ = build(yaml_name, dev, train=True, spread=True)
    ava = cfg.CONFIG.DATA.DATASET_NAME == "ava"
    clips = synth.structured_clips(len(sizes), 32, sizes[0][0], sizes[0][1], seed=int(gold["clip_seed"])).to(dev)
    targets = synth.synthetic_targets(len(sizes), "ava" if ava else "jhmdb", cfg.CONFIG.DATA.NUM_CLASSES, seed=int(gold["target_seed"]),
                                      hw=sizes[0], boxes_per_clip=[int(v) for v in gold["boxes_per_clip"]] if ava else None, device=dev)
    store, _ = model.engine()
    store.zero_grad()
    out = model(clips)
    got = flat_outputs(out)
    for k in ("pred_logits", "pred_boxes", "pred_logits_b"):
        print("%s train-mode output %-14s max abs err hip %.3e | bf16-rounded oracle %.3e" % (
            name, k, float(np.abs(got[k] - gold["out." + k]).max()), float(gold["rounded_err." + k])))
    criterion_probe(out)
    ld = crit(out, targets)
    wd = crit.weight_dict
    loss = sum(ld[k] * wd[k] for k in ld if k in wd)
    loss.backward()
    torch.cuda.synchronize()
    check_criterion_on_model_outputs(cfg, crit, out, targets, ld, tag=name)
    # the assignment against the REFERENCE's, per problem
    probs = matcher_problems(cfg, {k: v for k, v in out.items() if k != "_stacked"}, targets)
    same, total, flipped_layers, lines = 0, 0, set(), []
    for li, per in enumerate(crit.last_indices):
        for b, (qi, ti) in enumerate(per):
            total += 1
            a_ref = (gold["match.%d.%d.src" % (li, b)], gold["match.%d.%d.tgt" % (li, b)])
            C_ref, C_hip = gold["cost.%d.%d" % (li, b)], probs[li][b][0]
            ratio_ref, noise_ref = float(gold["ratio.%d.%d" % (li, b)]), float(gold["noise.%d.%d" % (li, b)])
            _, ratio_hip, noise_hip = assignment_margin(C_ref, a_ref, C_hip)
            ok = np.array_equal(qi.numpy(), a_ref[0]) and np.array_equal(ti.numpy(), a_ref[1])
            same += int(ok)
            lines.append("%d.%d %s margin %.2f ratio hip %.1f / rounded %.1f" % (li, b, "same" if ok else "FLIP", float(gold["margin.%d.%d" % (li, b)]), ratio_hip, ratio_ref))
            lines[-1] += " noise %.3f / %.3f" % (noise_hip, noise_ref)
            if not ok:
                flipped_layers.add(li)
                assert ratio_ref < 6.0, "decidable problem (layer %d, clip %d: gap >= 6 x the rounded oracle's perturbation) assigned differently from the reference" % (li, b)
                assert noise_hip <= 3.0 * noise_ref + 2e-2, "flip at (layer %d, clip %d) needs a cost perturbation of %.3f; the rounded oracle's is %.3f" % (li, b, noise_hip, noise_ref)
    print("%s matcher assignments identical to the reference: %d / %d   [%s]" % (name, same, total, "; ".join(lines)))
    print("%s query spread of the reference: boxes %s%s" % (name, gold["box_spread"][0].round(3), ", p_b in [%.3f, %.3f]" % tuple(gold["p_b_range"]) if ava else ""))
    assert same >= total - 1, (same, total)          # at most ONE flip of 12 (rounds 5-6: 12 / 11 / 11-12 of 12 over every build), and only under the conditions above
    worst_term = 0.0
    for k in sorted(ld):
        if k == "class_error":
            continue
        g, r = float(ld[k]), float(gold["loss." + k])
        layer = 0 if "_" not in k[5:] or not k.rsplit("_", 1)[1].isdigit() else int(k.rsplit("_", 1)[1]) + 1
        rel = abs(g - r) / max(abs(r), 1e-3)
        # yardstick: the bf16-rounded oracle's worst relative error on this loss FAMILY over the decoder layers (one layer's realised
        # error is a noisy estimate of the family's)
        fam = k if layer == 0 else k.rsplit("_", 1)[0]
        rel_rounded = max(abs(float(gold["rounded_loss." + kk]) - float(gold["loss." + kk])) / max(abs(float(gold["loss." + kk])), 1e-3)
                          for kk in ld if kk != "class_error" and (kk == fam or (kk.startswith(fam + "_") and kk[len(fam) + 1:].isdigit())))
        print("  %-14s hip %.5f  ref %.5f  (%.2f %%; bf16-rounded oracle %.2f %%)" % (k, g, r, 100 * rel, 100 * rel_rounded))
        if layer not in flipped_layers:
            worst_term = max(worst_term, rel)
            assert rel <= max(0.02, 4.0 * rel_rounded + 0.01), (k, g, r, rel_rounded)
    tl, tr = float(loss), float(gold["total_loss"])
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
    gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    print("%s total loss hip %.5f ref %.5f (%.3f %%); global grad norm hip %.4f ref %.4f (%.2f %%); worst loss term %.2f %%" % (
        name, tl, tr, 100 * abs(tl - tr) / abs(tr), gn, float(gold["grad_norm"]), 100 * abs(gn - float(gold["grad_norm"])) / float(gold["grad_norm"]), 100 * worst_term))
    names, norms = list(gold["grad_names"]), gold["grad_norms"]
    ratios = sorted((float(grads[str(n)].norm()) / rn, str(n)) for n, rn in zip(names, norms) if rn > 1e-3 * float(gold["grad_norm"]))
    print("  per-parameter grad-norm ratio hip/ref over %d significant tensors: min %.3f (%s)  median %.3f  max %.3f (%s)" % (
        len(ratios), ratios[0][0], ratios[0][1], ratios[len(ratios) // 2][0], ratios[-1][0], ratios[-1][1]))
    cos_low = []
    for k in gold.files:
        if k.startswith("grad."):
            n = k[5:]
            a, b = grads[n].flatten().double(), torch.as_tensor(gold[k]).flatten().double()
            cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
            print("  cos(grad %-44s) = %.4f   |hip| %.3e |ref| %.3e" % (n, cos, float(a.norm()), float(b.norm())))
            yard = float(gold["rounded_cos." + n])
            print("      noise-to-signal %.3f  (fully rounded oracle: cos %.4f, noise-to-signal %.3f)" % (_nsr(cos), yard, _nsr(yard)))
            if _nsr(cos) > 3.0 * _nsr(yard) + 0.3 or (n in ("class_embed_b.weight", "class_fc.bias") and cos < 0.98):
                cos_low.append((n, cos, yard))
    for k in gold.files:
        if k.startswith("buf."):
            v = dict(model.named_buffers())[k[4:]].float().cpu().numpy()
            assert float(np.abs(v - gold[k]).max()) <= 0.02 * max(1.0, float(np.abs(gold[k]).max())), k
    if not flipped_layers:
        y_tl = abs(float(gold["rounded_total_loss"]) - tr) / abs(tr)
        y_gn = abs(float(gold["rounded_grad_norm"]) - float(gold["grad_norm"])) / float(gold["grad_norm"])
        y_lo, y_hi = (float(v) for v in gold["rounded_ratio_range"])
        print("  fully rounded oracle on this fixture: total loss %.3f %%, global grad norm %.2f %%, per-tensor norm ratios [%.3f, %.3f]" % (
            100 * y_tl, 100 * y_gn, y_lo, y_hi))
        assert abs(tl - tr) <= max(0.005, 2.0 * y_tl + 0.001) * abs(tr)
        assert abs(gn - float(gold["grad_norm"])) <= max(0.05, 2.0 * y_gn + 0.01) * float(gold["grad_norm"])
        assert 0.67 * min(1.0, y_lo) < ratios[0][0] and ratios[-1][0] < 1.5 * max(1.0, y_hi), (ratios[0], ratios[-1])
        assert not cos_low, cos_low
    else:
        assert abs(tl - tr) <= 0.02 * abs(tr)


class _RoundBF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _surrogate(out):
    """smooth loss: fixed random linear functional of every output (no Hungarian discontinuity)."""
    g = torch.Generator().manual_seed(5)
    tot = 0
    for o in [out] + list(out.get("aux_outputs", [])):
        for k in ("pred_logits", "pred_boxes", "pred_logits_b"):
            tot = tot + (o[k].float() * torch.randn(o[k].shape, generator=g).to(o[k].device)).sum()
    return tot


@pytest.mark.parametrize("yaml_name,hw", [("TubeR_CSN152_AVA21.yaml", (64, 96)), ("TubeR_CSN50_AVA21.yaml", (64, 64)),
                                          ("Tuber_CSN152_JHMDB.yaml", (64, 64))])
def test_backward_vs_oracle(dev, yaml_name, hw):
    """Full forward+backward of the HIP model (shallow CSN-TEST body, every code path of the schedule: strided and
    stride-1 projection shortcuts, identity blocks, stem) against autograd of the fp32 oracle, with the accuracy of a
    bf16-ROUNDED oracle run as the yardstick: for every parameter, cos(hip, fp32) >= cos(bf16-rounded oracle, fp32) - 0.05
    (and >= 0.9 wherever the rounded oracle reaches 0.99)."""
    import torch.nn.functional as F
    from oracle import tuber_oracle as O
    cfg = load_cfg(os.path.join(ROOT, "configuration", yaml_name))
    cfg.CONFIG.MODEL.BACKBONE_NAME = "CSN-TEST"
    model, _, _ = build_model(cfg)
    synth.load_name_hashed(model)
    synth.zero_dropout(model)
    pn = [n for n, _ in model.named_parameters()]
    state = {k: v.clone() for k, v in model.state_dict().items()}
    clips = synth.synthetic_clips(2, 32, hw[0], hw[1], seed=99)

    def run_oracle(rounded):
        st = {k: (v.clone().requires_grad_(True) if k in pn else v.clone()) for k, v in state.items()}
        oc, ol = F.conv3d, F.linear
        if rounded:
            O.F.conv3d = lambda x, w, *a, **k: _RoundBF.apply(oc(_RoundBF.apply(x), _RoundBF.apply(w), *a, **k))
            O.F.linear = lambda x, w, b=None: _RoundBF.apply(ol(_RoundBF.apply(x), _RoundBF.apply(w), b))
        try:
            out = O.tuber_forward(st, cfg, clips, train=True)
            _surrogate(out).backward()
        finally:
            O.F.conv3d, O.F.linear = oc, ol
        return {k: st[k].grad for k in pn}, out

    g32, o32 = run_oracle(False)
    gbf, obf = run_oracle(True)
    model.to(dev).train()
    store, _ = model.engine()
    store.zero_grad()
    out = model(clips.to(dev))
    _surrogate(out).backward()
    torch.cuda.synchronize()
    for k in ("pred_logits", "pred_boxes", "pred_logits_b"):
        e_hip = float((out[k].float().cpu() - o32[k]).abs().max())
        e_bf = float((obf[k] - o32[k]).abs().max())
        print("train-mode %-14s |hip - fp32| %.3e   |bf16-rounded oracle - fp32| %.3e" % (k, e_hip, e_bf))
        assert e_hip <= 3 * e_bf + 2e-2
    worse, rows = [], []
    gnorm = math.sqrt(sum(float((g.double() ** 2).sum()) for g in g32.values() if g is not None))
    for n, p in model.named_parameters():
        a = g32[n]
        if a is None or float(a.norm()) < 1e-5 * gnorm:
            continue
        a = a.flatten().double()
        h = p.grad.detach().float().cpu().flatten().double()
        b = gbf[n].flatten().double()
        cb = float(a @ b / (a.norm() * b.norm() + 1e-30))
        if cb < 0.3:      # the fp32 gradient of this tensor is itself rounding noise (e.g. exactly-zero analytic gradient)
            continue
        ch = float(a @ h / (a.norm() * h.norm() + 1e-30))
        eh, eb = float((h - a).norm() / a.norm()), float((b - a).norm() / a.norm())
        nr = float(h.norm() / a.norm())
        rows.append((ch, cb, eh, eb, nr, n))
        if eh > 2.0 * eb + 0.05 or not (0.5 < nr < 2.0):
            worse.append((n, "cos %.4f/%.4f" % (ch, cb), "relerr %.3f/%.3f" % (eh, eb), "norm %.3f" % nr))
    rows.sort()
    med = len(rows) // 2
    print("parameters compared: %d; median cos hip %.4f (bf16-rounded oracle %.4f); median rel err hip %.4f (oracle %.4f)" % (
        len(rows), rows[med][0], sorted(r[1] for r in rows)[med], sorted(r[2] for r in rows)[med], sorted(r[3] for r in rows)[med]))
    for ch, cb, eh, eb, nr, n in rows[:10]:
        print("  lowest: %-56s cos hip %.4f  cos bf16-oracle %.4f  relerr %.3f / %.3f  norm ratio %.3f" % (n, ch, cb, eh, eb, nr))
    assert not worse, "gradients worse than 2x a bf16-rounded oracle: %s" % worse[:20]


@pytest.mark.gpu
def test_eval_loop_writes_reference_format_and_scores(tmp_path):
    """row N2 end to end on the GPU: validate_tuber_detection over a two-batch synthetic loader writes {rank}.txt / GT_{rank}.txt in
    the reference's format and returns a finite frame-mAP; the epoch training loop runs on the same loader."""
    from tubelet_transformer_amd.evaluation import validate_tuber_detection
    from tubelet_transformer_amd.training import build_optimizer, train_tuber_detection
    dev = torch.device("cuda:0")
    cfg = load_cfg(os.path.join(ROOT, "configuration", "TubeR_CSN50_AVA21.yaml"))
    cfg.CONFIG.LOG.BASE_PATH, cfg.CONFIG.LOG.RES_DIR = str(tmp_path), "tmp_res"
    model, criterion, post = build_model(cfg)
    synth.load_name_hashed(model)
    model.to(dev)
    criterion.to(dev)
    H, W = 64, 96
    loader = []
    for i in range(2):
        clips = synth.synthetic_clips(2, 32, H, W, seed=10 + i)
        tg = synth.synthetic_targets(2, "ava", 80, seed=20 + i, device="cpu", hw=(H, W))
        for b, t in enumerate(tg):
            n = t["boxes"].shape[0]
            t["image_id"] = ["vid%d_%04d" % (i, 900 + b), 16]
            t["size"] = torch.tensor([H, W])
            raw = torch.zeros(n, 6)
            raw[:, 0] = b                      # index of the clip inside the batch
            raw[:, 1] = 16                     # key-frame position
            raw[:, 2:] = torch.rand(n, 4).sort(dim=1).values[:, [0, 1, 2, 3]] * torch.tensor([W, H, W, H]) / 2 + torch.tensor([0, 0, W / 2, H / 2])
            t["raw_boxes"] = raw
        loader.append((clips, tg))
    mAP = validate_tuber_detection(cfg, model, criterion, post, loader, epoch=0, verbose=False)
    assert mAP == mAP and 0.0 <= mAP <= 1.0
    det = open(os.path.join(str(tmp_path), "tmp_res", "0.txt")).read().splitlines()
    gt = open(os.path.join(str(tmp_path), "tmp_res", "GT_0.txt")).read().splitlines()
    Q = cfg.CONFIG.MODEL.QUERY_NUM
    assert len(det) == 2 * 2 * Q
    key, rest = det[0].split(" [")
    assert key == "vid0_0900" and len(rest.split("]")[0].split(",")) == 4 + 80 + 1      # box, class scores, actor probability
    assert len(gt) == sum(t["boxes"].shape[0] for _, tg in loader for t in tg)
    assert len(gt[0].split(" [")[1].split("]")[0].split(",")) == 6 + 80
    # one training epoch over the same loader (eager steps)
    opt = build_optimizer(model, cfg)
    w0 = model.class_fc.weight.detach().clone()
    loader2 = [(c, [{k: v for k, v in t.items()} for t in tg]) for c, tg in loader]
    loss = train_tuber_detection(cfg, model, criterion, loader2, opt, epoch=0, max_norm=cfg.CONFIG.LOSS_COFS.CLIPS_MAX_NORM, print_freq=100)
    assert torch.isfinite(loss) and not torch.equal(w0, model.class_fc.weight.detach())


@pytest.mark.gpu
def test_split_graph_step_matches_single_graph(monkeypatch):
    """the DDP form of the captured step (hipGraph cut where layer3's backward ends, so the gradient all-reduce can run under
    layer2/layer1/stem) replays exactly the same kernels as the single graph: identical parameters after two steps"""
    from tubelet_transformer_amd.training import GraphedTrainStep, build_optimizer
    dev = torch.device("cuda:0")
    results = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("TUBER_FORCE_SPLIT_GRAPH", "1")
        else:
            monkeypatch.delenv("TUBER_FORCE_SPLIT_GRAPH", raising=False)
        cfg = load_cfg(os.path.join(ROOT, "configuration", "TubeR_CSN50_AVA21.yaml"))
        torch.manual_seed(0)
        model, criterion, _ = build_model(cfg)
        synth.load_name_hashed(model)
        model.to(dev).train()
        criterion.to(dev).train()
        opt = build_optimizer(model, cfg)
        clips = synth.synthetic_clips(2, 32, 64, 96, seed=3, device=dev)
        targets = synth.synthetic_targets(2, "ava", 80, seed=5, device=dev, hw=(64, 96))
        step = GraphedTrainStep(model, criterion, opt, cfg.CONFIG.LOSS_COFS.CLIPS_MAX_NORM)
        store, _ = model.engine()
        store.manual_seed(123)
        for _ in range(2):
            loss, _ = step(clips, targets)
        torch.cuda.synchronize()
        g = next(iter(step.graphs.values()))
        assert (g.A2 is not None) == force
        results.append((float(loss), store.flat.detach().clone()))
    assert results[0][0] == results[1][0]
    assert torch.equal(results[0][1], results[1][1])


@pytest.mark.gpu
def test_deferred_weight_gradient_reductions_are_bit_identical(monkeypatch):
    """one tuber_multi_reduce launch per backward pass instead of ~240 second-stage reductions: same summation order, so the captured
    step must produce exactly the parameters of the immediate path (TUBER_AB=immediate_reduce), eagerly and from the hipGraph."""
    from tubelet_transformer_amd.training import GraphedTrainStep, build_optimizer
    dev = torch.device("cuda:0")
    results = []
    from tubelet_transformer_amd import ab
    for immediate in (True, False):
        # (no_ln_bwd_fusion on both sides: the fused LayerNorm backward exists only with deferred reductions and sums the dgamma / dbeta rows in
        # 32-row blocks instead of 16 -- another, equally fixed order; this test is about the reductions)
        monkeypatch.setattr(ab, "_active", {"immediate_reduce", "no_ln_bwd_fusion"} if immediate else {"no_ln_bwd_fusion"})
        poison = [torch.full((1 << 28,), float("nan"), device=dev) for _ in range(6)]    # freed blocks the arena will be carved from:
        del poison                                                                        # a partial read before it is written shows as NaN
        cfg = load_cfg(os.path.join(ROOT, "configuration", "TubeR_CSN50_AVA21.yaml"))
        torch.manual_seed(0)
        model, criterion, _ = build_model(cfg)
        synth.load_name_hashed(model)
        model.to(dev).train()
        criterion.to(dev).train()
        opt = build_optimizer(model, cfg)
        clips = synth.synthetic_clips(2, 32, 128, 160, seed=3, device=dev)      # big enough that the dW GEMMs split into slabs
        targets = synth.synthetic_targets(2, "ava", 80, seed=5, device=dev, hw=(128, 160))
        step = GraphedTrainStep(model, criterion, opt, cfg.CONFIG.LOSS_COFS.CLIPS_MAX_NORM)
        store, _ = model.engine()
        assert store.defer.enabled == (not immediate)
        store.manual_seed(123)
        for _ in range(3):
            loss, _ = step(clips, targets)
        torch.cuda.synchronize()
        if not immediate:
            assert store.defer.cache and not store.defer.entries
            ntab = sum(int(v[0].numel()) // store.defer._ENTRY.itemsize for v in store.defer.cache.values())
            assert ntab > 50, "expected many deferred reductions, got %d" % ntab
        results.append((float(loss), store.flat.detach().clone(), store.gflat.detach().clone()))
    assert results[0][0] == results[1][0] and results[0][0] == results[0][0]
    assert torch.equal(results[0][2], results[1][2])
    assert torch.equal(results[0][1], results[1][1])


# ------------------------------------------------------------------ BASELINE.json full size (CSN-152, 3x32x256x340) ----------------------
@pytest.mark.gpu
def test_full_size_training_step_properties(dev):
    """size-independent properties of the whole fwd+bwd at BASELINE size (2 clips of 3x32x256x340, CSN-152, dropout on):
    (1) determinism: the same seed reproduces every gradient bit for bit;
    (2) linearity of the backward pass: doubling every loss weight doubles every gradient EXACTLY (powers of two commute with bf16 /
        fp32 rounding), which would break if any gradient buffer were read before being written or accumulated twice;
    (3) batch equivariance in eval mode: swapping the two clips swaps the outputs bit for bit (no cross-sample leakage)."""
    cfg, model, crit, _ = build("TubeR_CSN152_AVA21.yaml", dev, train=True)
    for m in model.modules():                                   # dropout back on (build() zeroes it for the parity cases)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.1
    store, _ = model.engine()
    clips = synth.synthetic_clips(2, 32, 256, 340, seed=1234, device=dev)
    targets = synth.synthetic_targets(2, "ava", 80, seed=4321, device=dev, hw=(256, 340))
    bn_state = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}

    def grads(scale):
        model.load_state_dict(bn_state, strict=False)           # same BatchNorm buffers for every run
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


@pytest.mark.gpu
def test_temporal_ds_strategy_max_matches_oracle(dev):
    """the fourth branch of Backbone.forward (backbone_builder.py:45-47,73; no published config uses it): eval forward against the
    oracle, and a training step whose stem gradient arrives through the max-pool routing"""
    from oracle import tuber_oracle as O
    cfg = load_cfg(os.path.join(ROOT, "configuration", "TubeR_CSN152_AVA21.yaml"))
    cfg.CONFIG.MODEL.TEMPORAL_DS_STRATEGY = "max"
    cfg.CONFIG.MODEL.BACKBONE_NAME = "CSN-TEST"
    model, crit, _ = build_model(cfg)
    synth.load_name_hashed(model)
    synth.zero_dropout(model)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev).eval()
    crit.to(dev)
    clips = synth.synthetic_clips(2, 32, 64, 96, seed=5)
    with torch.no_grad():
        want = flat_outputs(O.tuber_forward(state, cfg, clips, train=False))
        got = flat_outputs(model(clips.to(dev)))
    for k, v in want.items():
        tol = 1e-2 if k.endswith("pred_boxes") else 5e-2
        assert np.abs(got[k] - v).max() <= tol, (k, float(np.abs(got[k] - v).max()))
    model.train()
    crit.train()
    store, _ = model.engine()
    store.zero_grad()
    targets = synth.synthetic_targets(2, "ava", 80, seed=6, device=dev, hw=(64, 96))
    ld = crit(model(clips.to(dev)), targets)
    sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict).backward()
    g = model.backbone.body.conv1.weight.grad
    assert bool(torch.isfinite(store.gflat).all()) and float(g.abs().max()) > 0


@pytest.mark.gpu
def test_two_rank_bench_captures_the_graph_step(tmp_path):
    """N > 1 launch path of bench.py on a one-GPU box: two ranks share cuda:0 and all-reduce over gloo (TUBER_SHARE_GPU).  Checks the
    contract line and that the DDP step really runs from the captured hipGraph -- the eager warm-up (reducer hooks attached) and the
    captured backward order their deferred reductions differently, which once made the capture fall back to eager launches."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(TUBER_SHARE_GPU="1", TUBER_DIST_BACKEND="gloo")
    # the driver's form: `python bench.py --gpus N`, no launcher -- bench.py starts its own ranks (VERDICT r03 missing #2)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "64",
           "--width", "96", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "capture failed" not in r.stdout + r.stderr, (r.stdout + r.stderr)[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["launch_mode"] == "hipgraph" and j["config"]["global_batch"] == 4
    assert [l for l in r.stdout.splitlines() if l.strip()][-1] == line and j["timed_region_s"] > 0      # the JSON line is the job's last stdout line
    assert j["value"] > 0 and math.isfinite(j["final_loss"])
    # the diagnosability object of the N > 1 line: who carried the gradients, how much, in how many pieces
    c = j["comm"]
    assert c["world"] == 2 and "process group (gloo)" in c["transport"] and c["ranks_seen_by_rccl"] is None
    assert c["windows_per_step"] >= 2 and c["bytes_per_step"] > 4 * 40e6 and "graph A" in c["launch"]


def test_one_rank_of_a_multi_gpu_lease_minus_its_peers(tmp_path):
    """The exact process the first 8-GPU lease will start, minus the peers: ``bench.py --gpus 1`` under the
    environment ``spawn_ranks`` / torchrun give a rank (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) with TUBER_FORCE_DDP=1 ->
    torch's nccl process group of one rank, the OWN RCCL communicator next to it (one librccl in the process), the reducer's side stream
    and the cut-graph step.  The ``comm`` object must say so: RCCL itself reports 1 rank, the exposed wait was measured."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               TUBER_BENCH_SPAWNED="1", TUBER_FORCE_DDP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("TUBER_SHARE_GPU", "TUBER_DIST_BACKEND", "TUBER_NO_OWN_RCCL"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--height", "64", "--width", "96",
           "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["config"]["launch_mode"] == "hipgraph" and math.isfinite(j["final_loss"]) and j["value"] > 0
    c = j["comm"]
    assert "own RCCL communicator" in c["transport"], c
    assert c["world"] == 1 and c["ranks_seen_by_rccl"] == 1 and c["rccl_version"] > 0, c
    assert c["exposed_ms"] is not None and c["exposed_ms"] >= 0.0, c
    assert c["windows_per_step"] >= 2 and c["bytes_per_step"] > 4 * 40e6 and "graph A" in c["launch"], c
    print("one-rank launcher-shaped bench: comm = %s" % c)


@pytest.mark.gpu
def test_single_frame_false_matches_oracle(dev):
    """SINGLE_FRAME: False (backbone_builder.py:70,81-86; no published config uses it): no temporal pooling, the DETR encoder attends
    over all T' x h x w tokens with the 3-D positional encoding and the padding mask repeated over T'.  Eval forward against the
    oracle with the bf16-rounded-oracle yardstick, ragged batch; and a training step whose gradients reach the stem."""
    from parity_util import output_errors, run_oracle
    cfg = load_cfg(os.path.join(ROOT, "configuration", "TubeR_CSN152_AVA21.yaml"))
    cfg.CONFIG.MODEL.SINGLE_FRAME = False
    cfg.CONFIG.MODEL.BACKBONE_NAME = "CSN-TEST"
    model, crit, _ = build_model(cfg)
    synth.load_name_hashed(model)
    synth.zero_dropout(model)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev).eval()
    crit.to(dev)
    clips = synth.synthetic_clips(2, 32, 0, 0, seed=5, sizes=[(64, 96), (48, 80)])
    want, _ = run_oracle(cfg, state, clips, train=False)
    rnd, _ = run_oracle(cfg, state, clips, train=False, rounded=True)
    with torch.no_grad():
        got = model([c.to(dev) for c in clips])
    errs = output_errors(got, want, rnd)
    print("SINGLE_FRAME False: hip / rounded oracle vs fp32: %s" % {k: "%.2e / %.2e" % v for k, v in errs.items()})
    for kind, (eh, eb) in errs.items():
        assert eh <= (1e-2 if kind == "pred_boxes" else 5e-2) and eh <= 2 * eb + (1e-3 if kind == "pred_boxes" else 4e-3), (kind, eh, eb)
    model.train()
    crit.train()
    store, _ = model.engine()
    store.zero_grad()
    full = synth.synthetic_clips(2, 32, 64, 96, seed=6, device=dev)
    targets = synth.synthetic_targets(2, "ava", 80, seed=6, device=dev, hw=(64, 96))
    ld = crit(model(full), targets)
    sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict).backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(store.gflat).all()) and float(model.backbone.body.conv1.weight.grad.abs().max()) > 0
