"""Shared helpers of the GPU parity tests: the fp32 CPU oracle, a bf16-ROUNDED execution of the same oracle (every conv3d / linear
reads bf16-rounded operands and rounds its result -- the arithmetic a bf16 tensor-core path performs with exact accumulation), and
the per-parameter gradient comparison against that yardstick."""
import math
import os

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class RoundBF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class RoundBFG(torch.autograd.Function):
    """what a bf16-STORED activation does: the value is rounded forward, its gradient is rounded backward"""
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class RoundG(torch.autograd.Function):
    """identity forward, bf16-rounded gradient backward (the dS operand of the attention backward's MFMAs)"""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class rounded_convs:
    """context: the oracle's conv3d / linear read bf16-rounded operands and round their result (exact accumulation)"""

    def __enter__(self):
        from oracle import tuber_oracle as O
        self.O, self.oc, self.ol = O, O.F.conv3d, O.F.linear
        oc, ol = self.oc, self.ol
        O.F.conv3d = lambda x, w, *a, **k: RoundBF.apply(oc(RoundBF.apply(x), RoundBF.apply(w), *a, **k))
        O.F.linear = lambda x, w, b=None: RoundBF.apply(ol(RoundBF.apply(x), RoundBF.apply(w), b))
        return self

    def __exit__(self, *exc):
        self.O.F.conv3d, self.O.F.linear = self.oc, self.ol
        return False


def grad_row(h, a, b):
    """(cos hip, cos rounded, relerr hip, relerr rounded, norm ratio) of a HIP gradient h and a bf16-rounded-oracle gradient b against
    the fp32 truth a"""
    a = a.detach().flatten().double().cpu()
    h = h.detach().float().flatten().double().cpu()
    b = b.detach().flatten().double().cpu()
    na = float(a.norm()) + 1e-30
    return (float(a @ h / (na * (float(h.norm()) + 1e-30))), float(a @ b / (na * (float(b.norm()) + 1e-30))),
            float((h - a).norm() / na), float((b - a).norm() / na), float(h.norm() / na))


def surrogate(out):
    """smooth loss: fixed random linear functional of every output (no Hungarian discontinuity)."""
    g = torch.Generator().manual_seed(5)
    tot = 0
    for o in [out] + list(out.get("aux_outputs", [])):
        for k in ("pred_logits", "pred_boxes", "pred_logits_b"):
            tot = tot + (o[k].float() * torch.randn(o[k].shape, generator=g).to(o[k].device)).sum()
    return tot


def flat_outputs(out):
    d = {k: v.detach().float().cpu().numpy() for k, v in out.items() if k not in ("aux_outputs", "_stacked")}
    for i, a in enumerate(out.get("aux_outputs", [])):
        for k, v in a.items():
            d["aux%d.%s" % (i, k)] = v.detach().float().cpu().numpy()
    return d


def run_oracle(cfg, state, clips, train, rounded=False, param_names=None, loss=None, mask=None):
    """-> (outputs, {name: grad} or None).  ``state`` is cloned; with ``loss`` the parameters require grad and loss(out) is
    back-propagated.  ``rounded``: True = conv3d / linear read bf16-rounded operands and round their result (straight-through
    gradients); "full" = additionally the activation GRADIENTS entering and leaving every conv3d / linear are rounded, and so are the
    attention probabilities and the gradient of the attention scores -- every place the HIP path stores or feeds an MFMA in bf16."""
    from oracle import tuber_oracle as O
    pn = set(param_names or [])
    st = {k: (v.clone().requires_grad_(True) if (loss is not None and k in pn) else v.clone()) for k, v in state.items()}
    oc, ol, osm = F.conv3d, F.linear, F.softmax
    if rounded == "full":
        O.F.conv3d = lambda x, w, *a, **k: RoundBFG.apply(oc(RoundBFG.apply(x), RoundBF.apply(w), *a, **k))
        O.F.linear = lambda x, w, b=None: RoundBFG.apply(ol(RoundBFG.apply(x), RoundBF.apply(w), b))
        O.F.softmax = lambda s, dim=-1, **k: RoundBFG.apply(osm(RoundG.apply(s), dim, **k))
    elif rounded:
        O.F.conv3d = lambda x, w, *a, **k: RoundBF.apply(oc(RoundBF.apply(x), RoundBF.apply(w), *a, **k))
        O.F.linear = lambda x, w, b=None: RoundBF.apply(ol(RoundBF.apply(x), RoundBF.apply(w), b))
    try:
        if loss is None:
            with torch.no_grad():
                out = O.tuber_forward(st, cfg, clips, mask=mask, train=train)
            return out, None
        out = O.tuber_forward(st, cfg, clips, mask=mask, train=train)
        loss(out).backward()
    finally:
        O.F.conv3d, O.F.linear, O.F.softmax = oc, ol, osm
    return out, {k: st[k].grad for k in pn}


def output_errors(got, want, rounded):
    """max abs error per output kind: {kind: (hip vs fp32, bf16-rounded oracle vs fp32)}"""
    import numpy as np
    g, w, r = flat_outputs(got), flat_outputs(want), flat_outputs(rounded)
    res = {}
    for k, v in w.items():
        assert g[k].shape == v.shape and np.isfinite(g[k]).all(), k
        kind = k.split(".")[-1]
        a, b = res.get(kind, (0.0, 0.0))
        res[kind] = (max(a, float(np.abs(g[k] - v).max())), max(b, float(np.abs(r[k] - v).max())))
    return res


def compare_gradients(named_hip_grads, g32, gbf, k=2.0, slack=0.05, min_cb=0.3):
    """per parameter: relative L2 error of the HIP gradient against fp32 autograd of the oracle must be <= k x the error of the
    bf16-rounded oracle + slack, norm ratio in (0.5, 2).  Tensors whose fp32 gradient is numerically zero are skipped, and --
    unless ``min_cb`` is None -- those for which the bf16-rounded oracle itself decorrelates from fp32 (cos < min_cb).
    -> (rows sorted by cosine: (cos_hip, cos_rounded, relerr_hip, relerr_rounded, norm ratio, name), offenders)"""
    gnorm = math.sqrt(sum(float((g.double() ** 2).sum()) for g in g32.values() if g is not None))
    rows, worse = [], []
    for n, h in named_hip_grads:
        a = g32.get(n)
        if a is None or h is None or float(a.norm()) < 1e-5 * gnorm:
            continue
        a = a.flatten().double()
        h = h.detach().float().cpu().flatten().double()
        b = gbf[n].flatten().double()
        cb = float(a @ b / (a.norm() * b.norm() + 1e-30))
        if min_cb is not None and cb < min_cb:
            continue
        ch = float(a @ h / (a.norm() * h.norm() + 1e-30))
        eh, eb = float((h - a).norm() / a.norm()), float((b - a).norm() / a.norm())
        nr = float(h.norm() / a.norm())
        rows.append((ch, cb, eh, eb, nr, n))
        if eh > k * eb + slack or not 0.5 < nr < 2.0:
            worse.append((n, "cos %.4f/%.4f" % (ch, cb), "relerr %.3f/%.3f" % (eh, eb), "norm %.3f" % nr))
    rows.sort()
    return rows, worse


def report(rows, tag):
    med = len(rows) // 2
    msg = "%s: parameters compared %d; median cos hip %.4f (bf16-rounded oracle %.4f); median rel err hip %.4f (oracle %.4f); worst hip rel err %.3f" % (
        tag, len(rows), rows[med][0], sorted(r[1] for r in rows)[med], sorted(r[2] for r in rows)[med], sorted(r[3] for r in rows)[med],
        max(r[2] for r in rows))
    print(msg)
    for ch, cb, eh, eb, nr, n in rows[:8]:
        print("  lowest: %-56s cos hip %.4f  cos bf16-oracle %.4f  relerr %.3f / %.3f  norm ratio %.3f" % (n, ch, cb, eh, eb, nr))
    return msg


def host_mem_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def criterion_probe(out):
    """Call BEFORE ``criterion(out, targets)`` / ``backward``: keep the gradients w.r.t. the model's outputs (non-leaf tensors of
    the one autograd node the network is) so that ``check_criterion_on_model_outputs`` can compare them."""
    ts = list(out["_stacked"]) if "_stacked" in out else \
        [o[k] for o in [out] + list(out.get("aux_outputs", [])) for k in ("pred_logits", "pred_logits_b", "pred_boxes")]
    for t in ts:
        if t.requires_grad:
            t.retain_grad()
    return ts


def check_criterion_on_model_outputs(cfg, crit, out, targets, ld, tag=""):
    """The fused HIP criterion fed by REAL model outputs against the oracle's criterion (the reference's arithmetic:
    models/detr/matcher.py:61-80, models/criterion.py:42-206 / :237-410) evaluated on THE SAME output values (copied to the host):

    * Hungarian assignment: identical on every (decoder layer, clip) -- integer work, bit-exact, no tolerance.  (Only if SciPy on the
      fp32 host cost and the device solver on its own fp32 cost disagree is an exact cost tie accepted: both assignments must then have
      the same total cost to 1e-6 on the host matrix.)
    * every loss term <= 1e-4 relative (floor 1e-6), class_error <= 1e-3;
    * gradient of the weighted total w.r.t. every output tensor <= 2e-5 abs (needs ``criterion_probe(out)`` before the backward).

    This is what makes the end-to-end train-step tests falsifiable for the matcher -> gather -> loss chain: a wrong index, a wrong
    gather or a wrong weight in the fused criterion fails here whatever bf16 noise the network put into ``out``.
    -> number of (layer, clip) problems checked"""
    import numpy as np
    from oracle import tuber_oracle as O

    def cpu_leaf(v):
        return v.detach().float().cpu().clone().requires_grad_(True)
    o = {k: cpu_leaf(v) for k, v in out.items() if k not in ("aux_outputs", "_stacked")}
    o["aux_outputs"] = [{k: cpu_leaf(v) for k, v in a.items()} for a in out.get("aux_outputs", [])]
    tg = [{k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]
    mld, idx = O.set_criterion(cfg, o, tg)
    total = O.total_loss(cfg, mld)
    total.backward()
    got = crit.last_indices
    assert len(got) == len(idx)
    problems = 0
    layers = [o] + o["aux_outputs"]
    for li, (per_h, per_o) in enumerate(zip(got, idx)):
        for b, ((qi, ti), (qo, to)) in enumerate(zip(per_h, per_o)):
            problems += 1
            if np.array_equal(qi.numpy(), qo.numpy()) and np.array_equal(ti.numpy(), to.numpy()):
                continue
            # exact tie?  both assignments on the host cost matrix of this problem
            lay = layers[li]
            if cfg.CONFIG.DATA.DATASET_NAME != "ava":
                nq = cfg.CONFIG.MODEL.QUERY_NUM
                kf = torch.stack([nq * t["key_pos"] + torch.arange(nq) for t in tg])
                lay = {k: (v.gather(1, kf[:, :, None].repeat(1, 1, v.shape[-1])) if k in ("pred_boxes", "pred_logits") else v) for k, v in lay.items()}
            _, C = O.hungarian_match(cfg, {k: v.detach() for k, v in lay.items()}, tg)
            sizes = [len(t["boxes"]) for t in tg]
            Cb = C.split(sizes, -1)[b][b].double()
            ch, co = float(Cb[qi, ti].sum()), float(Cb[qo, to].sum())
            assert abs(ch - co) <= 1e-6, "%s layer %d clip %d: device assignment %s/%s (cost %.7f) vs SciPy on the same outputs %s/%s (cost %.7f)" % (
                tag, li, b, qi.tolist(), ti.tolist(), ch, qo.tolist(), to.tolist(), co)
    worst = 0.0
    for k, v in mld.items():
        g, r = float(ld[k]), float(v)
        worst = max(worst, abs(g - r) / max(abs(r), 1e-6) if abs(r) > 1e-6 else abs(g - r))
        assert abs(g - r) <= 1e-4 * abs(r) + 1e-6, (tag, k, g, r)
    gw = None
    if "_stacked" in out and all(t.grad is not None for t in out["_stacked"] if t.requires_grad):
        gw = 0.0
        lay_order = o["aux_outputs"] + [o]                        # decoder-layer order of the stacked tensors
        for t, key in zip(out["_stacked"], ("pred_logits", "pred_logits_b", "pred_boxes")):
            if not t.requires_grad or t.grad is None:
                continue
            hip = t.grad.detach().float().cpu()
            for l, lay in enumerate(lay_order):
                ref = lay[key].grad if lay[key].grad is not None else torch.zeros_like(lay[key])
                if hip[l].shape != ref.shape:                     # JHMDB logits_b is [B,2] per layer, replicated: compare what exists
                    continue
                gw = max(gw, float((hip[l] - ref).abs().max()))
        assert gw <= 2e-5, (tag, "gradient w.r.t. the outputs", gw)
    print("%s criterion on the model's own outputs vs the oracle's criterion on the same values: %d / %d assignments identical, worst loss-term "
          "relative error %.2e, worst output-gradient abs error %s" % (tag, problems, problems, worst, "%.2e" % gw if gw is not None else "n/a"))
    return problems


# ---- Hungarian-assignment decidability (round 5: the non-degenerate "spread" fixtures) --------------------------------------------
def matcher_problems(cfg, out, targets):
    """The matcher's cost matrices (models/detr/matcher.py:61-80 via the oracle) of every (decoder layer, clip) problem for the
    outputs ``out`` (host tensors): list over layers in the reference's order (main, aux_0 ..) of lists over clips of
    (C float32 [Q, n_targets], (query_idx, target_idx)) -- JHMDB with the key-frame gather of criterion.py:378-380 applied."""
    from oracle import tuber_oracle as O
    nq = cfg.CONFIG.MODEL.QUERY_NUM
    ava = cfg.CONFIG.DATA.DATASET_NAME == "ava"
    layers = [{k: v for k, v in out.items() if k not in ("aux_outputs", "_stacked")}] + list(out.get("aux_outputs", []))
    sizes = [len(t["boxes"]) for t in targets]
    res = []
    for o in layers:
        o = {k: v.detach().float().cpu() for k, v in o.items()}
        if not ava:
            kf = torch.stack([nq * t["key_pos"].cpu() + torch.arange(nq) for t in targets])
            o = {k: (v.gather(1, kf[:, :, None].repeat(1, 1, v.shape[-1])) if k in ("pred_boxes", "pred_logits") else v) for k, v in o.items()}
        idx, C = O.hungarian_match(cfg, o, [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets])
        res.append([(c[i].numpy(), (idx[i][0].numpy(), idx[i][1].numpy())) for i, c in enumerate(C.split(sizes, -1))])
    return res


def assignment_margin(C_ref, a_ref, C_other=None):
    """For one problem: (margin, ratio, noise).  margin = cost of the second-best assignment - cost of the optimal one ``a_ref`` on ``C_ref``
    (exhaustive over the injective maps targets -> queries: n <= 3, Q <= 15 in the fixtures).  With ``C_other`` (the same problem's
    costs from a perturbed execution, e.g. the bf16-rounded oracle): ratio = min over the alternatives a of
    gap_ref(a) / (|gap_ref(a) - gap_other(a)| + 1e-3) -- how many times the realised perturbation of an alternative's gap fits into
    the gap.  ratio >> 1: the assignment is decidable under that noise; ratio <~ 1: any execution with such noise may flip it.
    noise = the largest such perturbation |gap_ref(a) - gap_other(a)| over the alternatives within 4 x margin + 0.5 of the optimum
    (the ones that matter)."""
    import itertools
    Q, n = C_ref.shape
    star = [None] * n
    for q, j in zip(a_ref[0], a_ref[1]):
        star[int(j)] = int(q)
    b1 = sum(C_ref[q, j] for j, q in enumerate(star))
    b2 = sum(C_other[q, j] for j, q in enumerate(star)) if C_other is not None else 0.0
    margin, ratio = float("inf"), float("inf")
    gaps = []
    for perm in itertools.permutations(range(Q), n):
        if list(perm) == star:
            continue
        g1 = float(sum(C_ref[q, j] for j, q in enumerate(perm)) - b1)
        margin = min(margin, g1)
        if C_other is not None:
            g2 = float(sum(C_other[q, j] for j, q in enumerate(perm)) - b2)
            ratio = min(ratio, g1 / (abs(g1 - g2) + 1e-3))
            gaps.append((g1, abs(g1 - g2)))
    noise = max([d for g, d in gaps if g <= 4 * margin + 0.5], default=0.0)
    return margin, ratio, noise
