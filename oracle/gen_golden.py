"""Pin the oracle against the imported reference and write tests/golden/*.npz.

Run in the BUILD CONTAINER only (needs /root/reference):  python oracle/gen_golden.py
For every case it (1) runs the unmodified reference (oracle/ref_import.py) with
name-hashed weights and seeded inputs, (2) runs oracle/tuber_oracle.py on the same
state dict, (3) asserts max|diff| <= 1e-5 (grads: 1e-4 relative to the grad scale), and
(4) stores the REFERENCE's outputs as the golden vector.  Weights and inputs are NOT
stored: tests regenerate them from names/seeds (tubelet_transformer_amd/synth.py).
"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, tuber_oracle as O                      # noqa: E402
from tubelet_transformer_amd import synth                             # noqa: E402
from tubelet_transformer_amd.config import load_cfg                   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
LOG = []

# name -> (yaml, list of clip (h,w), extra)
MODEL_CASES = {
    "csn50_ava21_decode_eval": ("TubeR_CSN50_AVA21.yaml", [(64, 96)]),
    "csn152_ava21_avg_eval_ragged": ("TubeR_CSN152_AVA21.yaml", [(64, 96), (48, 80)]),
    "csn152_ava22_decode_eval": ("TubeR_CSN152_AVA22.yaml", [(64, 64)]),
    "csn152_jhmdb_eval": ("Tuber_CSN152_JHMDB.yaml", [(64, 64)]),
}
TRAIN_CASES = {
    "csn152_ava21_avg_train": ("TubeR_CSN152_AVA21.yaml", [(64, 96), (64, 96)]),
    "csn50_ava21_decode_train": ("TubeR_CSN50_AVA21.yaml", [(64, 64), (64, 64)]),
    "csn152_jhmdb_train": ("Tuber_CSN152_JHMDB.yaml", [(64, 64), (64, 64)]),
}


# round 5: non-degenerate ("spread") train fixtures -- synth.SPREAD_GAINS + residual_gain 0.05 + structured clips; the clip / target
# seeds are chosen by spread_search below so that the reference's Hungarian assignment is as decidable under bf16 noise as it gets
SPREAD_TRAIN_CASES = {
    "csn152_ava21_avg_train_spread": ("TubeR_CSN152_AVA21.yaml", [(64, 96), (64, 96)]),
    "csn50_ava21_decode_train_spread": ("TubeR_CSN50_AVA21.yaml", [(64, 64), (64, 64)]),
    "csn152_jhmdb_train_spread": ("Tuber_CSN152_JHMDB.yaml", [(64, 64), (64, 64)]),
}
SPREAD_EVAL_CASES = {
    "csn152_ava21_avg_eval_spread": ("TubeR_CSN152_AVA21.yaml", [(64, 96), (64, 96)]),
}
SPREAD_RESIDUAL_GAIN = 0.05
SPREAD_BOXES_PER_CLIP = [1, 2]


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s)
    LOG.append(s)


def flat_outputs(out):
    d = {k: v.detach().numpy() for k, v in out.items() if k != "aux_outputs"}
    for i, a in enumerate(out.get("aux_outputs", [])):
        for k, v in a.items():
            d["aux%d.%s" % (i, k)] = v.detach().numpy()
    return d


def maxdiff(a, b):
    return max(float(np.abs(a[k] - b[k]).max()) for k in a)


def make_clips(sizes, seed):
    if len(set(sizes)) == 1:
        return synth.synthetic_clips(len(sizes), 32, sizes[0][0], sizes[0][1], seed=seed)
    return synth.synthetic_clips(len(sizes), 32, 0, 0, seed=seed, sizes=sizes)


def eval_case(name, yaml_name, sizes, spread=False):
    rcfg = ref_import.ref_cfg(yaml_name)
    mycfg = load_cfg(os.path.join(ROOT, "configuration", yaml_name))
    model, _, post = ref_import.build_reference(rcfg)
    if spread:
        synth.load_name_hashed(model, residual_gain=SPREAD_RESIDUAL_GAIN, spread=True)
    else:
        synth.load_name_hashed(model)
    model.eval()
    clips = synth.structured_clips(len(sizes), 32, sizes[0][0], sizes[0][1], seed=1234) if spread else make_clips(sizes, seed=1234)
    with torch.no_grad():
        ref = model(clips)
        state = {k: v.clone() for k, v in model.state_dict().items()}
        mine = O.tuber_forward(state, mycfg, clips, train=False)
    r, m = flat_outputs(ref), flat_outputs(mine)
    d = maxdiff(r, m)
    log("[eval ] %-32s oracle-vs-reference max|diff| = %.3e" % (name, d))
    assert d <= (5e-5 if spread else 1e-5), name
    # post-processing on the same outputs
    tsz = torch.tensor([[h * 4, w * 4] for h, w in sizes], dtype=torch.int64)
    pr = post["bbox"](ref, tsz)
    pm = O.post_process(mycfg, {k: v for k, v in mine.items() if k != "aux_outputs"}, tsz)
    dp = max(float(np.abs(a - b).max()) / max(1.0, float(np.abs(a).max())) for a, b in zip(pr, pm))
    log("        %-32s post-process max rel diff (boxes are in pixels) = %.3e" % ("", dp))
    assert dp <= 1e-5
    r.update({"post.scores": pr[0], "post.boxes": pr[1], "post.out_b": pr[2], "post.target_sizes": tsz.numpy()})
    if spread and mycfg.CONFIG.DATA.DATASET_NAME == "ava":
        pb = ref["pred_logits_b"].softmax(-1)[..., 1]
        b = ref["pred_boxes"]
        log("        %-32s eval spread: p_b in [%.3f, %.3f] (%d of %d queries above the 0.8 gate), box spread (clip 0) %s" % (
            "", float(pb.min()), float(pb.max()), int((pb > 0.8).sum()), pb.numel(), (b.max(1).values - b.min(1).values)[0].numpy().round(3)))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **r)


# (clip seed, target seed) spread_search chose for each spread fixture (GENERATION_LOG.txt)
SPREAD_CHOSEN = {"csn152_ava21_avg_train_spread": (99, 181), "csn50_ava21_decode_train_spread": (101, 94), "csn152_jhmdb_train_spread": (100, 25)}


def spread_search(mycfg, state, sizes, ava, clip_seeds=(99, 100, 101, 102), target_seeds=range(200)):
    """Choose (clip seed, target seed) for a spread fixture: the pair for which the fp32 assignment of every (layer, clip) problem is
    the most decidable under the noise of a bf16-ROUNDED execution of the oracle (tests/parity_util.assignment_margin: worst ratio
    gap / realised gap perturbation over all alternatives of all problems), among those the rounded oracle assigns like fp32."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_util import run_oracle, matcher_problems, assignment_margin
    best = None
    for cs in clip_seeds:
        clips = synth.structured_clips(len(sizes), 32, sizes[0][0], sizes[0][1], seed=cs)
        o32, _ = run_oracle(mycfg, state, clips, train=True, rounded=False)
        obf, _ = run_oracle(mycfg, state, clips, train=True, rounded=True)
        for ts in target_seeds:
            tg = synth.synthetic_targets(len(sizes), "ava" if ava else "jhmdb", mycfg.CONFIG.DATA.NUM_CLASSES, seed=ts, hw=sizes[0],
                                         boxes_per_clip=SPREAD_BOXES_PER_CLIP if ava else None)
            p32, pbf = matcher_problems(mycfg, o32, tg), matcher_problems(mycfg, obf, tg)
            ratios, same = [], True
            for la, lb in zip(p32, pbf):
                for (C1, a1), (C2, a2) in zip(la, lb):
                    ratios.append(assignment_margin(C1, a1, C2)[1])
                    same = same and np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1], a2[1])
            if same and (best is None or min(ratios) > best[0]):
                best = (min(ratios), cs, ts, sorted(ratios))
    return best


def train_case(name, yaml_name, sizes, spread=False):
    rcfg = ref_import.ref_cfg(yaml_name)
    mycfg = load_cfg(os.path.join(ROOT, "configuration", yaml_name))
    # the published JHMDB yaml ships EVAL_ONLY: True; keep it (it only affects the AVA loss)
    ava = mycfg.CONFIG.DATA.DATASET_NAME == "ava"
    model, crit, _ = ref_import.build_reference(rcfg)
    if spread:
        synth.load_name_hashed(model, residual_gain=SPREAD_RESIDUAL_GAIN, spread=True)
    else:
        synth.load_name_hashed(model)
    ref_import.zero_dropout(model)
    model.train()
    crit.train()
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    if spread and name in SPREAD_CHOSEN and "research" not in sys.argv:
        # the seeds an earlier spread_search chose (pass "research" to search again); the decidability ratios are recomputed below
        clip_seed, target_seed = SPREAD_CHOSEN[name]
        log("[spread] %-30s pinned clip seed %d, target seed %d" % (name, clip_seed, target_seed))
        clips = synth.structured_clips(len(sizes), 32, sizes[0][0], sizes[0][1], seed=clip_seed)
        targets = synth.synthetic_targets(len(sizes), "ava" if ava else "jhmdb", mycfg.CONFIG.DATA.NUM_CLASSES,
                                          seed=target_seed, hw=sizes[0], boxes_per_clip=SPREAD_BOXES_PER_CLIP if ava else None)
    elif spread:
        worst, clip_seed, target_seed, ratios = spread_search(mycfg, state0, sizes, ava)
        log("[spread] %-30s chosen clip seed %d, target seed %d: worst decidability ratio (gap / bf16-rounded-oracle perturbation) %.2f; "
            "per problem %s" % (name, clip_seed, target_seed, worst, " ".join("%.1f" % r for r in ratios)))
        clips = synth.structured_clips(len(sizes), 32, sizes[0][0], sizes[0][1], seed=clip_seed)
        targets = synth.synthetic_targets(len(sizes), "ava" if ava else "jhmdb", mycfg.CONFIG.DATA.NUM_CLASSES,
                                          seed=target_seed, hw=sizes[0], boxes_per_clip=SPREAD_BOXES_PER_CLIP if ava else None)
    else:
        clips = make_clips(sizes, seed=99)
        targets = synth.synthetic_targets(len(sizes), "ava" if ava else "jhmdb", mycfg.CONFIG.DATA.NUM_CLASSES,
                                          seed=7, hw=sizes[0], boxes_per_clip=[2, 3] if ava else None)
    out = model(clips)
    ld = crit(out, targets)
    wd = crit.weight_dict
    loss = sum(ld[k] * wd[k] for k in ld if k in wd)
    loss.backward()
    ref_grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    ref_losses = {k: float(v) for k, v in ld.items() if k != "class_error"}
    ref_state1 = model.state_dict()

    # oracle on the same initial state
    pnames = {n for n, _ in model.named_parameters()}
    st = {k: (v.clone().requires_grad_(True) if k in pnames else v.clone()) for k, v in state0.items()}
    mo = O.tuber_forward(st, mycfg, clips, train=True)
    mld, idx = O.set_criterion(mycfg, mo, targets)
    mloss = O.total_loss(mycfg, mld)
    mloss.backward()
    d_out = maxdiff(flat_outputs(out), flat_outputs(mo))
    d_loss = max(abs(ref_losses[k] - float(mld[k])) for k in ref_losses)
    log("[train] %-32s outputs %.3e  losses %.3e  total %.6f vs %.6f" % (name, d_out, d_loss, float(loss), float(mloss)))
    # spread fixtures: head gains x2 / x3 and sharper attention scale the fp32 summation-order noise of the outputs with them
    assert d_out <= (5e-5 if spread else 1e-5) and d_loss <= 1e-4 * max(1.0, abs(float(loss)))
    worst = 0.0
    for n, g in ref_grads.items():
        og = st[n].grad
        assert og is not None, n
        # scale: the param's own grad magnitude, floored (params whose true grad is ~0, e.g. the q/k rows of a
        # 1-key softmax in the LSTR pool decoder, carry only rounding noise)
        scale = max(float(g.abs().max()), 1e-3 if spread else 1e-4)
        rel = float((g - og).abs().max()) / scale
        if rel > 1e-3:
            log("        note: %s rel %.3e (|g|max %.3e)" % (n, rel, float(g.abs().max())))
        worst = max(worst, rel)
    log("        %-32s worst per-param relative grad diff = %.3e over %d params" % ("", worst, len(ref_grads)))
    assert worst <= 2e-3
    missing = [n for n in pnames if n not in ref_grads]
    log("        params with no grad in the reference:", missing)
    # BN buffers after the step
    d_buf = max(float((ref_state1[k].float() - st[k].detach().float()).abs().max()) for k in state0 if k not in pnames)
    log("        %-32s buffers after step max|diff| = %.3e" % ("", d_buf))
    assert d_buf <= 1e-5

    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in ref_grads.values()))
    names = sorted(ref_grads)
    keep = ["backbone.body.conv1.weight", "backbone.body.bn1.weight", "backbone.body.layer1.0.conv3.weight",
            "backbone.body.layer4.2.bn4.bias", "query_embed.weight", "class_embed_b.weight", "class_fc.bias",
            "transformer.decoder.norm.weight", "encoder.layers.0.norm1_s.weight", "input_proj.bias"]
    gold = {"loss." + k: np.float64(v) for k, v in ref_losses.items()}
    gold["total_loss"] = np.float64(float(loss))
    gold["grad_norm"] = np.float64(float(gn))
    gold["grad_names"] = np.array(names)
    gold["grad_norms"] = np.array([float(ref_grads[n].double().norm()) for n in names])
    for k in keep:
        if k in ref_grads:
            gold["grad." + k] = ref_grads[k].numpy()
    for li, ind in enumerate(idx):
        for b, (i, j) in enumerate(ind):
            gold["match.%d.%d.src" % (li, b)] = i.numpy()
            gold["match.%d.%d.tgt" % (li, b)] = j.numpy()
    for k in ("backbone.body.bn1.running_mean", "backbone.body.bn1.running_var",
              "backbone.body.layer3.1.bn3.running_var", "backbone.body.layer4.0.down_sample.1.running_mean"):
        gold["buf." + k] = ref_state1[k].numpy()
    gold.update({"out." + k: v for k, v in flat_outputs(out).items()})
    if spread:
        # what the test needs to judge decidability per problem: the reference's cost matrices, and how far a bf16-rounded execution
        # of the oracle moves every alternative's gap (ratio); the spread actually reached
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from parity_util import run_oracle, matcher_problems, assignment_margin
        obf, _ = run_oracle(mycfg, state0, clips, train=True, rounded=True)
        pref = matcher_problems(mycfg, {k: v for k, v in out.items()}, targets)
        pbf = matcher_problems(mycfg, obf, targets)
        for li, (la, lb) in enumerate(zip(pref, pbf)):
            for b, ((C1, a1), (C2, a2)) in enumerate(zip(la, lb)):
                assert np.array_equal(a1[0], gold["match.%d.%d.src" % (li, b)]) and np.array_equal(a1[1], gold["match.%d.%d.tgt" % (li, b)])
                assert np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1], a2[1]), "the bf16-rounded oracle must keep the reference's assignment"
                m, r, nz = assignment_margin(C1, a1, C2)
                gold["cost.%d.%d" % (li, b)] = C1
                gold["margin.%d.%d" % (li, b)] = np.float64(m)
                gold["ratio.%d.%d" % (li, b)] = np.float64(r)
                gold["noise.%d.%d" % (li, b)] = np.float64(nz)
        gold["clip_seed"], gold["target_seed"] = np.int64(clip_seed), np.int64(target_seed)
        gold["boxes_per_clip"] = np.array(SPREAD_BOXES_PER_CLIP if ava else [1] * len(sizes))
        b = out["pred_boxes"].detach()
        gold["box_spread"] = (b.max(1).values - b.min(1).values).numpy()
        if ava:
            pb = out["pred_logits_b"].detach().softmax(-1)[..., 1]
            gold["p_b_range"] = np.array([float(pb.min()), float(pb.max())])
            log("        %-32s box spread over the queries (clip 0) %s, p_b in [%.3f, %.3f]" % ("", gold["box_spread"][0].round(3), float(pb.min()), float(pb.max())))
        for k in ("pred_logits", "pred_boxes", "pred_logits_b"):
            gold["rounded_err." + k] = np.float64(float((out[k].detach() - obf[k]).abs().max()))
        # the loss terms of the bf16-rounded execution (same assignment as the reference, asserted above): the yardstick of the test's
        # per-term tolerance
        rld, _ = O.set_criterion(mycfg, obf, targets)
        for k, v in rld.items():
            gold["rounded_loss." + k] = np.float64(float(v))
        # ... and the GRADIENTS of the fully rounded execution (parity_util.run_oracle rounded="full": also the activation gradients,
        # the attention probabilities and the score gradients go through bf16, as they do in the HIP path; exact accumulation): how
        # far ideal bf16 arithmetic moves the total loss, the global gradient norm, each stored tensor's direction and the per-tensor
        # norms on this fixture -- the yardstick of the test's gradient tolerances
        full = {}

        def full_loss(o):
            fld, fidx = O.set_criterion(mycfg, o, targets)
            full["loss"], full["same"] = float(O.total_loss(mycfg, fld)), all(
                np.array_equal(i1.numpy(), i2.numpy()) and np.array_equal(j1.numpy(), j2.numpy())
                for la, lb in zip(idx, fidx) for (i1, j1), (i2, j2) in zip(la, lb))
            return O.total_loss(mycfg, fld)
        _, gbf = run_oracle(mycfg, state0, clips, train=True, rounded="full", param_names=sorted(ref_grads), loss=full_loss)
        assert full["same"], "the fully rounded oracle must keep the reference's assignment"
        gold["rounded_total_loss"] = np.float64(full["loss"])
        gold["rounded_grad_norm"] = np.float64(math.sqrt(sum(float((g.double() ** 2).sum()) for g in gbf.values() if g is not None)))
        rr = []
        for n in names:
            a, b = gbf[n].flatten().double(), ref_grads[n].flatten().double()
            if n in keep:
                gold["rounded_cos." + n] = np.float64(float((a @ b) / (a.norm() * b.norm() + 1e-30)))
            if float(b.norm()) > 1e-3 * float(gn):
                rr.append(float(a.norm() / b.norm()))
        gold["rounded_ratio_range"] = np.array([min(rr), max(rr)])
        log("        %-32s bf16-rounded oracle: total loss %.3f %%, grad norm %.2f %%, per-tensor norm ratios [%.3f, %.3f], cosines %s" % (
            "", 100 * abs(float(gold["rounded_total_loss"]) - float(loss)) / abs(float(loss)),
            100 * abs(float(gold["rounded_grad_norm"]) - float(gn)) / float(gn), min(rr), max(rr),
            " ".join("%s=%.3f" % (n.split(".")[-2][-8:] + "." + n.split(".")[-1][0], float(gold["rounded_cos." + n])) for n in keep if n in ref_grads)))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **gold)


def criterion_case(name, yaml_name, seed):
    """Fixed outputs dict + targets -> all loss values, matcher indices, grads wrt outputs."""
    rcfg = ref_import.ref_cfg(yaml_name)
    mycfg = load_cfg(os.path.join(ROOT, "configuration", yaml_name))
    ava = mycfg.CONFIG.DATA.DATASET_NAME == "ava"
    nc, nq = mycfg.CONFIG.DATA.NUM_CLASSES, mycfg.CONFIG.MODEL.QUERY_NUM
    _, crit, _ = ref_import.build_reference(rcfg)
    g = torch.Generator().manual_seed(seed)
    bs = 3
    Q = nq if ava else nq * 32

    def mk():
        o = {"pred_logits": torch.randn(bs, Q, nc if ava else nc + 1, generator=g),
             "pred_boxes": torch.rand(bs, Q, 4, generator=g) * 0.5 + 0.25,
             "pred_logits_b": torch.randn(bs, Q, 3, generator=g) if ava else torch.randn(bs, 2, generator=g)}
        return o
    outs = mk()
    outs["aux_outputs"] = [mk() for _ in range(5)]
    targets = synth.synthetic_targets(bs, "ava" if ava else "jhmdb", nc, seed=seed + 1,
                                      boxes_per_clip=[1, 4, 2] if ava else None)

    def run(fn):
        leaves = {}

        def req(o, pfx):
            r = {}
            for k, v in o.items():
                r[k] = v.clone().requires_grad_(True)
                leaves[pfx + k] = r[k]
            return r
        o = req({k: v for k, v in outs.items() if k != "aux_outputs"}, "")
        o["aux_outputs"] = [req(a, "aux%d." % i) for i, a in enumerate(outs["aux_outputs"])]
        ld, total = fn(o)
        total.backward()
        return ld, total, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}

    def ref_fn(o):
        ld = crit(o, targets)
        wd = crit.weight_dict
        return ld, sum(ld[k] * wd[k] for k in ld if k in wd)

    def my_fn(o):
        ld, idx = O.set_criterion(mycfg, o, targets)
        my_fn.idx = idx
        return ld, O.total_loss(mycfg, ld)

    rl, rt, rg = run(ref_fn)
    ml, mt, mg = run(my_fn)
    dl = max(abs(float(rl[k]) - float(ml[k])) for k in ml)
    dg = max(float((rg[k] - mg[k]).abs().max()) for k in rg)
    log("[crit ] %-32s losses %.3e  grads %.3e  total %.6f" % (name, dl, dg, float(rt)))
    assert dl <= 1e-5 and dg <= 1e-5
    gold = {"loss." + k: np.float64(float(v)) for k, v in rl.items() if k != "class_error"}
    gold["class_error"] = np.float64(float(rl["class_error"]))
    gold["total_loss"] = np.float64(float(rt))
    for k, v in rg.items():
        gold["grad." + k] = v.numpy()
    for k, v in outs.items():
        if k != "aux_outputs":
            gold["in." + k] = v.numpy()
    for i, a in enumerate(outs["aux_outputs"]):
        for k, v in a.items():
            gold["in.aux%d.%s" % (i, k)] = v.numpy()
    for li, ind in enumerate(my_fn.idx):
        for b, (i, j) in enumerate(ind):
            gold["match.%d.%d.src" % (li, b)] = i.numpy()
            gold["match.%d.%d.tgt" % (li, b)] = j.numpy()
    gold["seed"] = np.int64(seed)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **gold)


def main():
    assert ref_import.available(), "run in the build container: /root/reference is required"
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    which = sys.argv[1:] or ["crit", "eval", "train", "spread"]
    if "crit" in which:
        criterion_case("criterion_ava", "TubeR_CSN152_AVA21.yaml", 11)
        criterion_case("criterion_jhmdb", "Tuber_CSN152_JHMDB.yaml", 12)
    if "eval" in which:
        for n, (y, s) in MODEL_CASES.items():
            eval_case(n, y, s)
    if "train" in which:
        for n, (y, s) in TRAIN_CASES.items():
            train_case(n, y, s)
    if "spread" in which:
        for n, (y, s) in SPREAD_TRAIN_CASES.items():
            train_case(n, y, s, spread=True)
        for n, (y, s) in SPREAD_EVAL_CASES.items():
            eval_case(n, y, s, spread=True)
    with open(os.path.join(GOLD, "GENERATION_LOG.txt"), "a") as f:
        f.write("\n".join(LOG) + "\n")
    json.dump({"torch": torch.__version__, "numpy": np.__version__}, open(os.path.join(GOLD, "versions.json"), "w"))


if __name__ == "__main__":
    main()


def state_dict_golden():
    """tests/golden/reference_state_dicts.json: key names + shapes of the reference's model / criterion state_dict and its
    weight_dict for every published config (the checkpoint-compatibility contract)."""
    out = {}
    for y in ["TubeR_CSN152_AVA21", "TubeR_CSN50_AVA21", "Tuber_CSN152_JHMDB", "TubeR_CSN152_AVA22"]:
        rm, rc, _ = ref_import.build_reference(ref_import.ref_cfg(y + ".yaml"))
        out[y] = {"model": [[k, list(v.shape)] for k, v in rm.state_dict().items()],
                  "criterion": [[k, list(v.shape)] for k, v in rc.state_dict().items()],
                  "weight_dict": {k: float(v) for k, v in rc.weight_dict.items()}}
    json.dump(out, open(os.path.join(GOLD, "reference_state_dicts.json"), "w"))
