"""Hungarian-matched set criterion and post-processing of TubeR on the MI355X path.

API mirror of ``models/criterion.py`` (``SetCriterionAVA`` :11-206, ``SetCriterion`` :209-410, ``PostProcess`` :413-445,
``PostProcessAVA`` :447-482) and ``models/detr/matcher{,_ucf}.py`` (``HungarianMatcher`` :37-81 / :37-88): same constructors,
``criterion(outputs, targets) -> dict`` with the 24 (+class_error) keys, mutable ``criterion.weight_dict``.

How it runs (SURVEY.md sections 2.3 K14/K15, 7.8):
  * ``tuber_criterion_cost``: matching cost of ALL decoder layers in one kernel, ONE device->host copy per step (the reference
    does six ``.cpu()`` round trips), assignment by ``tuber_lsap`` -- the C++ restatement of SciPy's solver in libtuber_hip.so;
  * ``tuber_criterion_loss``: every loss term of every layer AND its gradient w.r.t. the model outputs in one launch, fp32;
    the autograd backward only scales those gradients by the loss weights.  The weighted BCE is evaluated from logits
    (softplus), equal to ``F.binary_cross_entropy(sigmoid(x), t, w)`` including its log clamp at -100 (criterion.py:57,71-73);
  * targets are padded to a static [B, Tmax] layout so the whole step can be captured in a hipGraph.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops, lib


def _lsap(cost):
    """cost: float64 numpy [nr, nc] -> (row_ind, col_ind) int64, identical to scipy.optimize.linear_sum_assignment."""
    L = lib.load()
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    k = min(nr, nc)
    ri, ci = np.zeros(k, np.int64), np.zeros(k, np.int64)
    if k == 0:
        return ri, ci
    rc = L.tuber_lsap(cost.ctypes.data, nr, nc, ri.ctypes.data, ci.ctypes.data)
    if rc != 0:
        raise ValueError("cost matrix is infeasible or contains invalid entries (tuber_lsap rc=%d)" % rc)
    return ri, ci


class PaddedTargets:
    """Static [B, Tmax] device layout of the target dicts."""

    def __init__(self, targets, ava, num_classes, device, tmax=None):
        self.sizes = [int(t["boxes"].shape[0]) for t in targets]
        B = len(targets)
        need = max(self.sizes + [1])
        self.tmax = tmax if tmax is not None else max(8, (need + 7) // 8 * 8)
        if need > self.tmax:
            raise ValueError("%d targets in a clip exceed Tmax=%d" % (need, self.tmax))
        self.ava, self.B = ava, B
        self.tboxes = torch.zeros(B, self.tmax, 4, dtype=torch.float32, device=device)
        self.tlabels = torch.zeros((B, self.tmax, num_classes) if ava else (B, self.tmax), dtype=torch.float32, device=device)
        self.tcount = torch.tensor(self.sizes, dtype=torch.int32).to(device)
        # JHMDB / UCF: key-frame position and visibility label per clip (criterion.py:378-380,256-262) -- device buffers too, so a
        # captured step reads the CURRENT batch's values on every replay
        self.key_pos = None if ava else torch.zeros(B, dtype=torch.int64, device=device)
        self.vis = None if ava else torch.zeros(B, dtype=torch.int64, device=device)
        self.fill(targets)

    def fill(self, targets):
        for b, t in enumerate(targets):
            n = self.sizes[b]
            if n:
                self.tboxes[b, :n] = t["boxes"][:, 1:].to(self.tboxes)          # column 0 is the key-frame index
                self.tlabels[b, :n] = t["labels"].to(self.tlabels)
        if not self.ava:
            if any(torch.as_tensor(t["vis"]).numel() != 1 for t in targets):
                raise ValueError("one visibility label per clip expected (datasets/jhmdb_frame.py:170-189)")
            self.key_pos.copy_(torch.stack([torch.as_tensor(t["key_pos"]).reshape(()) for t in targets]).to(self.key_pos), non_blocking=True)
            self.vis.copy_(torch.stack([torch.as_tensor(t["vis"]).reshape(()) for t in targets]).to(self.vis), non_blocking=True)

    def refill(self, targets):
        """new batch into the same device buffers (hipGraph replays read these addresses)."""
        sizes = [int(t["boxes"].shape[0]) for t in targets]
        if len(sizes) != self.B or max(sizes + [0]) > self.tmax:
            raise ValueError("batch of %d clips with up to %d targets does not fit the captured [%d, %d] layout" % (len(sizes), max(sizes + [0]), self.B, self.tmax))
        self.sizes = sizes
        if self._pack(targets):
            return
        self.tboxes.zero_()
        self.tlabels.zero_()
        self.tcount.copy_(torch.tensor(sizes, dtype=torch.int32), non_blocking=True)
        self.fill(targets)

    def _pack(self, targets):
        """the whole refill as ONE launch (tuber_targets_pack) when every per-clip tensor already sits on the device in the layout the
        reference's collate gives it (boxes fp32 [n,5], labels fp32 [n,C] / int64 [n]); False -> the per-clip copies above."""
        dev = self.tboxes.device
        if dev.type != "cuda" or self.B > lib.query("tuber_targets_pack_max"):
            return False
        ldt = torch.float32 if self.ava else torch.int64
        C = self.tlabels.shape[2] if self.ava else 1
        for t, n in zip(targets, self.sizes):
            bx, lb = t["boxes"], t["labels"]
            if not (torch.is_tensor(bx) and torch.is_tensor(lb) and bx.device == dev and lb.device == dev and bx.dtype == torch.float32 and lb.dtype == ldt
                    and bx.is_contiguous() and lb.is_contiguous() and tuple(bx.shape) == (n, 5) and tuple(lb.shape) == ((n, C) if self.ava else (n,))):
                return False
        import ctypes
        P = ctypes.c_void_p * self.B
        boxes = P(*[t["boxes"].data_ptr() if n else None for t, n in zip(targets, self.sizes)])
        labels = P(*[t["labels"].data_ptr() if n else None for t, n in zip(targets, self.sizes)])
        sizes = (ctypes.c_int * self.B)(*self.sizes)
        lib.call("tuber_targets_pack", boxes, labels, sizes, self.B, self.tmax, C, 1 if self.ava else 0, self.tboxes, self.tlabels, self.tcount)
        if not self.ava:
            self.key_pos.copy_(torch.stack([torch.as_tensor(t["key_pos"]).reshape(()) for t in targets]).to(self.key_pos), non_blocking=True)
            self.vis.copy_(torch.stack([torch.as_tensor(t["vis"]).reshape(()) for t in targets]).to(self.vis), non_blocking=True)
        return True


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class=1.0, cost_bbox=1.0, cost_giou=1.0, data_file="ava", binary_loss=False, before=False):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.data_file, self.binary_loss, self.before = data_file, binary_loss, before
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def cost(self, logits, logits_b, boxes, pt):
        """[L,B,Q,Tmax] cost tensor on the device (entries beyond tcount[b] are 0)."""
        L, B, Q, C = logits.shape
        out = torch.empty(L, B, Q, pt.tmax, dtype=torch.float32, device=logits.device)
        lib.call("tuber_criterion_cost", logits, logits_b, boxes, pt.tboxes, pt.tlabels, pt.tcount, L, B, Q, C, pt.tmax,
                 1 if self.data_file == "ava" else 0, float(self.cost_class), float(self.cost_bbox), float(self.cost_giou), out)
        return out

    @staticmethod
    def solve(cost_host, sizes):
        """cost_host numpy [L,B,Q,Tmax] -> (match int32 [L,B,Tmax] (query of target j or -1), indices[l][b] = (idx_q, idx_t))."""
        L, B, Q, T = cost_host.shape
        match = np.full((L, B, T), -1, dtype=np.int32)
        indices = []
        for l in range(L):
            per = []
            for b, n in enumerate(sizes):
                i, j = _lsap(cost_host[l, b, :, :n].astype(np.float64))
                match[l, b, j] = i
                per.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
            indices.append(per)
        return match, indices

    @torch.no_grad()
    def forward(self, outputs, targets):
        ava = self.data_file == "ava"
        lg = outputs["pred_logits"].float().contiguous()[None]
        bx = outputs["pred_boxes"].float().contiguous()[None]
        lb = outputs["pred_logits_b"].float().contiguous()[None] if ava else lg
        pt = PaddedTargets(targets, ava, lg.shape[-1], lg.device)
        C = self.cost(lg, lb, bx, pt).cpu().numpy()
        return self.solve(C, pt.sizes)[1][0]


def build_matcher(cfg):
    M = cfg.CONFIG.MATCHER
    return HungarianMatcher(cost_class=M.COST_CLASS, cost_bbox=M.COST_BBOX, cost_giou=M.COST_GIOU,
                            data_file=cfg.CONFIG.DATA.DATASET_NAME, binary_loss=M.BNY_LOSS, before=M.BEFORE)


def _class_error(logits, pt, match, ava):
    B, Q, C = logits.shape
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    lib.call("tuber_class_error", logits.contiguous(), match.contiguous(), pt.tlabels, B, Q, C, pt.tmax, 1 if ava else 0, out)
    return out[0]


class _LossFn(torch.autograd.Function):
    """losses[L,4] = (ce, ce_b, bbox, giou) per decoder layer; gradients precomputed by the forward kernel."""

    @staticmethod
    def forward(ctx, logits, logits_b, boxes, pt, match, ava, eos, pos_weight):
        L, B, Q, C = logits.shape
        dev = logits.device
        losses = torch.empty(L, 4, dtype=torch.float32, device=dev)
        g_l = torch.empty_like(logits)
        g_b = torch.empty(L, B, Q, 3, dtype=torch.float32, device=dev)
        g_x, g_g = torch.empty_like(boxes), torch.empty_like(boxes)
        lib.call("tuber_criterion_loss", logits, logits_b, boxes, pt.tboxes, pt.tlabels, pt.tcount, match, L, B, Q, C, pt.tmax,
                 1 if ava else 0, float(eos), float(pos_weight), losses, g_l, g_b, g_x, g_g)
        ctx.save_for_backward(g_l, g_b, g_x, g_g)
        ctx.ava = ava
        return losses

    @staticmethod
    def backward(ctx, g):
        g_l, g_b, g_x, g_g = ctx.saved_tensors
        g = g.float().contiguous()
        L = g_l.shape[0]
        gl, gx = torch.empty_like(g_l), torch.empty_like(g_x)
        gb = torch.empty_like(g_b) if ctx.ava else None
        lib.call("tuber_criterion_scale", g, g_l, g_b if ctx.ava else None, g_x, g_g, L, g_l.numel() // L, g_b.numel() // L, g_x.numel() // L, gl, gb, gx)
        return gl, gb, gx, None, None, None, None, None


class _WeightedSum(torch.autograd.Function):
    """sum(lv * W) of the [L,4] loss table and the [L,4] weight table in one launch each way (tuber_weighted_sum)"""

    @staticmethod
    def forward(ctx, lv, W):
        lv = lv.contiguous()
        out = torch.empty((), dtype=torch.float32, device=lv.device)
        lib.call("tuber_weighted_sum", lv, W, lv.numel(), out, None, None)
        ctx.save_for_backward(W)
        ctx.shape = lv.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (W,) = ctx.saved_tensors
        gl = torch.empty(ctx.shape, dtype=torch.float32, device=W.device)
        lib.call("tuber_weighted_sum", None, W, W.numel(), None, g.float().contiguous(), gl)
        return gl, None


class _SetCriterionBase(nn.Module):
    def __init__(self, weight, num_classes, num_queries, matcher, weight_dict, eos_coef, losses, data_file, evaluation=False):
        super().__init__()
        self.weight, self.evaluation = weight, evaluation
        self.num_classes, self.num_queries = num_classes, num_queries
        self.matcher, self.weight_dict = matcher, weight_dict
        self.eos_coef, self.losses, self.data_file = eos_coef, losses, data_file
        self.ava = data_file == "ava"
        self._indices, self._match_dev = None, None

    # -- assignment ---------------------------------------------------------------------------------------
    @property
    def last_indices(self):
        """reference order (main, aux_0 .. aux_4) list of per-clip (query_idx, target_idx) pairs of the last call; when the
        assignment ran on the device this is where it is copied to the host (lazily: training never needs it)."""
        if self._indices is None and self._match_dev is not None:
            match, sizes = self._match_dev
            m = match.cpu().numpy()
            L = m.shape[0]
            per_layer = []
            for l in range(L):
                per = []
                for b, n in enumerate(sizes):
                    q = m[l, b, :n]
                    t = np.nonzero(q >= 0)[0]
                    order = np.argsort(q[t], kind="stable")
                    per.append((torch.as_tensor(q[t][order], dtype=torch.int64), torch.as_tensor(t[order], dtype=torch.int64)))
                per_layer.append(per)
            self._indices = [per_layer[L - 1]] + per_layer[:L - 1]
        return self._indices

    @last_indices.setter
    def last_indices(self, v):
        self._indices, self._match_dev = v, None

    def assign(self, cost, pt):
        """cost [L,B,Q,Tmax] (device) -> match int32 [L,B,Tmax] on the device.  tuber_lsap_device when the problems fit its
        128 x 128 bound (no host round trip), else the host tuber_lsap."""
        L, B, Q, T = cost.shape
        if Q <= 128 and T <= 128:
            match = torch.empty(L, B, T, dtype=torch.int32, device=cost.device)
            lib.call("tuber_lsap_device", cost, pt.tcount, match, L, B, Q, T)
            self._indices, self._match_dev = None, (match, list(pt.sizes))
            return match
        m, indices = self.matcher.solve(cost.cpu().numpy(), pt.sizes)
        self.last_indices = [indices[L - 1]] + indices[:L - 1]
        return torch.from_numpy(m).to(cost.device, non_blocking=True)

    # -- pieces (also used one by one by the hipGraph-captured step, training.GraphedStep) ----------------
    def stacked(self, outputs):
        """(logits, logits_b, boxes) as [L,B,Q,.] fp32 in decoder-layer order (main output = last layer)."""
        if "_stacked" in outputs:
            return outputs["_stacked"]
        layers = list(outputs.get("aux_outputs", [])) + [{k: v for k, v in outputs.items() if k != "aux_outputs"}]
        return tuple(torch.stack([o[k].float() for o in layers]) for k in ("pred_logits", "pred_logits_b", "pred_boxes"))

    def select(self, logits, boxes, pt):
        return logits, boxes

    def losses_from_match(self, logits, logits_b, boxes, pt, match_dev, targets=None):
        L = logits.shape[0]
        pos_w = 1.0 if (self.evaluation or not self.ava) else float(self.weight)
        lv = _LossFn.apply(logits.contiguous(), logits_b.contiguous() if self.ava else logits, boxes.contiguous(), pt, match_dev,
                           self.ava, float(self.eos_coef), pos_w)
        ce_b = lv[:, 1] if self.ava else self.visibility_loss(logits_b, pt)
        out = {}
        for l in range(L):
            sfx = "" if l == L - 1 else "_%d" % l
            out["loss_ce" + sfx] = lv[l, 0]
            out["loss_ce_b" + sfx] = ce_b[l]
            out["loss_bbox" + sfx] = lv[l, 2]
            out["loss_giou" + sfx] = lv[l, 3]
        self._lv = (lv, ce_b if not self.ava else None)       # for weighted_total: one fused weighted sum instead of 24 scalar ops
        return out

    def weighted_total(self, loss_dict, weight_dict=None):
        """sum_k weight_dict[k] * loss_dict[k] (train_tuber_ava.py / video_action_recognition.py:147) -- computed as ONE weighted
        reduction over the stacked [L,4] loss tensor when ``loss_dict`` is the dict this criterion just returned (the 24 scalar
        multiplies / adds and their autograd nodes cost ~190 tiny launches per step otherwise)."""
        wd = self.weight_dict if weight_dict is None else weight_dict
        lv_pack = getattr(self, "_lv", None)
        if lv_pack is None or loss_dict.get("loss_ce") is None or loss_dict["loss_ce"]._base is not lv_pack[0]:
            return sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
        lv, ce_b = lv_pack
        L = lv.shape[0]
        names = ("loss_ce", "loss_ce_b", "loss_bbox", "loss_giou")
        vals = tuple(float(wd.get(n + ("" if l == L - 1 else "_%d" % l), 0.0)) for l in range(L) for n in names)
        W, wb = self.sync_weights(lv.device, vals)
        if ce_b is None:
            return _WeightedSum.apply(lv, W)
        Wm = W.clone()
        Wm[:, 1] = 0
        total = (lv * Wm).sum() + (ce_b * wb).sum()
        return total

    def sync_weights(self, device, vals=None):
        """the [L,4] loss-weight tensor on the device, refreshed IN PLACE from ``weight_dict`` when a value changed (the
        ``epoch > WEIGHT_CHANGE`` switch of loss_ce, video_action_recognition.py:145-146) -- a captured hipGraph keeps reading the
        same address, so the change takes effect on the next replay.  Call it before replaying; inside a capture nothing is copied."""
        bufs = self.__dict__.setdefault("_w_bufs", {})
        key = str(device)
        ent = bufs.get(key)
        if vals is None:
            if ent is None:
                return None
            L = ent[0].shape[0]
            names = ("loss_ce", "loss_ce_b", "loss_bbox", "loss_giou")
            vals = tuple(float(self.weight_dict.get(n + ("" if l == L - 1 else "_%d" % l), 0.0)) for l in range(L) for n in names)
        if ent is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("loss-weight buffer must exist before hipGraph capture (run one eager step first)")
            W = torch.tensor(vals, dtype=torch.float32).view(-1, 4).to(device)
            ent = bufs[key] = [W, W[:, 1].clone(), vals]
        elif ent[2] != vals and not torch.cuda.is_current_stream_capturing():
            W = torch.tensor(vals, dtype=torch.float32).view(-1, 4)
            ent[0].copy_(W)
            ent[1].copy_(W[:, 1])
            ent[2] = vals
        return ent[0], ent[1]

    def forward(self, outputs, targets):
        logits, logits_b, boxes = self.stacked(outputs)
        pt = PaddedTargets(targets, self.ava, logits.shape[-1], logits.device)
        logits_s, boxes_s = self.select(logits, boxes, pt)
        with torch.no_grad():
            cost = self.matcher.cost(logits_s.detach().contiguous(), (logits_b if self.ava else logits_s).detach().contiguous(),
                                     boxes_s.detach().contiguous(), pt)
            match_dev = self.assign(cost, pt)
        losses = self.losses_from_match(logits_s, logits_b, boxes_s, pt, match_dev)
        losses["class_error"] = self.class_error(logits_s[-1], pt, match_dev[-1])
        return losses


class SetCriterionAVA(_SetCriterionBase):
    """AVA: 3-way actor CE (target 1 matched / 2 unmatched, weights [1,1,eos]) + weighted multi-label BCE (criterion.py:42-81)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        ew = torch.ones(3)
        ew[-1] = self.eos_coef
        self.register_buffer("empty_weight", ew)

    @torch.no_grad()
    def class_error(self, logits, pt, match):
        """100 - exact-set accuracy of the matched queries (utils/misc.py:497-518): top-k(labels) == labels  <=>  min logit over the
        labels > max logit over the rest.  One kernel, no sync."""
        return _class_error(logits, pt, match, True)


class SetCriterion(_SetCriterionBase):
    """JHMDB/UCF: (C+1)-way CE with no-object weight, 2-way visibility CE, key-frame query gather (criterion.py:237-262,378-396)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        ew = torch.ones(self.num_classes + 1)
        ew[-1] = self.eos_coef
        self.register_buffer("empty_weight", ew)

    def select(self, logits, boxes, pt):
        nq = self.num_queries
        dev = logits.device
        kf = nq * pt.key_pos[:, None] + torch.arange(nq, device=dev)[None, :]                                # [B,nq]
        L = logits.shape[0]
        idx = kf[None, :, :, None].expand(L, -1, -1, -1)
        return (torch.gather(logits, 2, idx.expand(-1, -1, -1, logits.shape[-1])),
                torch.gather(boxes, 2, idx.expand(-1, -1, -1, 4)))

    def visibility_loss(self, logits_b, pt):
        vis = pt.vis
        L, B = logits_b.shape[:2]
        return F.cross_entropy(logits_b.reshape(L * B, -1).float(), vis.repeat(L), reduction="none").view(L, B).mean(1)

    @torch.no_grad()
    def class_error(self, logits, pt, match):
        """100 - top-1 accuracy of the matched queries (utils/misc.py:521-539)."""
        return _class_error(logits, pt, match, False)


class PostProcess(nn.Module):
    @torch.no_grad()
    def forward(self, outputs, target_sizes):
        lg, bx, lb = outputs["pred_logits"].float(), outputs["pred_boxes"].float(), outputs["pred_logits_b"].float()
        assert len(lg) == len(target_sizes) and target_sizes.shape[1] == 2
        prob = F.softmax(lg, -1)
        h, w = target_sizes.to(bx.device).unbind(1)
        boxes = box_ops.box_cxcywh_to_xyxy(bx) * torch.stack([w, h, w, h], dim=1)[:, None, :]
        return prob.cpu().numpy(), boxes.cpu().numpy(), lb.softmax(-1).cpu().numpy()[..., 1:]


class PostProcessAVA(nn.Module):
    @torch.no_grad()
    def forward(self, outputs, target_sizes):
        lb, lg, bx = outputs["pred_logits_b"].float(), outputs["pred_logits"].float(), outputs["pred_boxes"].float()
        assert len(lg) == len(target_sizes) and target_sizes.shape[1] == 2
        pb = lb.softmax(-1)[:, :, 1:2]
        prob = lg.sigmoid() * ((pb > 0.8).float() * pb)
        h, w = target_sizes.to(bx.device).unbind(1)
        boxes = box_ops.box_cxcywh_to_xyxy(bx) * torch.stack([w, h, w, h], dim=1)[:, None, :]
        return prob.cpu().numpy(), boxes.cpu().numpy(), pb.cpu().numpy()
