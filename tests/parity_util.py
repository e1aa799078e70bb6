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
    back-propagated."""
    from oracle import tuber_oracle as O
    pn = set(param_names or [])
    st = {k: (v.clone().requires_grad_(True) if (loss is not None and k in pn) else v.clone()) for k, v in state.items()}
    oc, ol = F.conv3d, F.linear
    if rounded:
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
        O.F.conv3d, O.F.linear = oc, ol
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
