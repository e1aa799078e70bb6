"""Hungarian-matched set criterion and post-processing of TubeR.

API mirror of ``models/criterion.py`` (``SetCriterionAVA`` :11-206, ``SetCriterion`` :209-410, ``PostProcess`` :413-445,
``PostProcessAVA`` :447-482) and ``models/detr/matcher{,_ucf}.py`` (``HungarianMatcher`` :37-81 / :37-88).

Differences in HOW (not what) it computes, chosen for the MI355X step (SURVEY.md sections 2.3 K14/K15, 7.8):
  * the matching cost of ALL decoder layers is built in one batched device computation and copied to the host ONCE
    per step (the reference does 6 ``.cpu()`` syncs), the assignment runs in the C++ restatement of SciPy's solver
    (``tuber_lsap`` in libtuber_hip.so -- no SciPy on the product path);
  * the weighted BCE is evaluated from logits in fp32 (``softplus``), which equals
    ``F.binary_cross_entropy(sigmoid(x), t, w)`` up to that function's log clamp at -100 (criterion.py:57,71-73).
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops, lib
from .misc import accuracy, accuracy_sigmoid


def _lsap(cost):
    """cost: float64 numpy [nr, nc] -> (row_ind, col_ind) int64, identical to scipy.optimize.linear_sum_assignment."""
    L = lib.load()
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    k = min(nr, nc)
    ri, ci = np.zeros(k, np.int64), np.zeros(k, np.int64)
    rc = L.tuber_lsap(cost.ctypes.data, nr, nc, ri.ctypes.data, ci.ctypes.data)
    if rc != 0:
        raise ValueError("cost matrix is infeasible or contains invalid entries (tuber_lsap rc=%d)" % rc)
    return ri, ci


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class=1.0, cost_bbox=1.0, cost_giou=1.0, data_file="ava", binary_loss=False, before=False):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.data_file, self.binary_loss, self.before = data_file, binary_loss, before
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def cost_matrices(self, layer_outputs, targets):
        """Cost tensors for a list of per-layer output dicts: [n_layers, B, Q, sum_i N_i] on the device."""
        ava = self.data_file == "ava"
        boxes = torch.stack([o["pred_boxes"] for o in layer_outputs]).float()           # [L,B,Q,4]
        Lr, B, Q, _ = boxes.shape
        ob = boxes.reshape(-1, 4)
        tb = torch.cat([t["boxes"] for t in targets])[:, 1:].float()
        c_bbox = torch.cdist(ob, tb, p=1)
        c_giou = -box_ops.generalized_box_iou(box_ops.box_cxcywh_to_xyxy(ob), box_ops.box_cxcywh_to_xyxy(tb))
        if ava:
            prob = torch.stack([o["pred_logits_b"] for o in layer_outputs]).float().reshape(Lr * B * Q, -1).softmax(-1)
            c_cls = -prob[:, 1:2].expand(-1, tb.shape[0])
        else:
            ids = torch.cat([t["labels"] for t in targets])
            prob = torch.stack([o["pred_logits"] for o in layer_outputs]).float().reshape(Lr * B * Q, -1).softmax(-1)
            c_cls = -prob[:, ids]
        C = self.cost_bbox * c_bbox + self.cost_class * c_cls + self.cost_giou * c_giou
        return C.view(Lr, B, Q, -1)

    @torch.no_grad()
    def match_layers(self, layer_outputs, targets):
        """indices[layer][b] = (idx_query int64, idx_target int64) CPU tensors; ONE device->host copy."""
        C = self.cost_matrices(layer_outputs, targets).cpu().double().numpy()
        sizes = [len(t["boxes"]) for t in targets]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        out = []
        for l in range(C.shape[0]):
            per = []
            for b, n in enumerate(sizes):
                i, j = _lsap(C[l, b, :, offs[b]:offs[b] + n])
                per.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
            out.append(per)
        return out

    @torch.no_grad()
    def forward(self, outputs, targets):
        return self.match_layers([outputs], targets)[0]


def build_matcher(cfg):
    M = cfg.CONFIG.MATCHER
    return HungarianMatcher(cost_class=M.COST_CLASS, cost_bbox=M.COST_BBOX, cost_giou=M.COST_GIOU,
                            data_file=cfg.CONFIG.DATA.DATASET_NAME, binary_loss=M.BNY_LOSS, before=M.BEFORE)


def _src_idx(indices, device):
    b = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)]).to(device)
    return b, torch.cat([s for s, _ in indices]).to(device)


class _SetCriterionBase(nn.Module):
    def __init__(self, weight, num_classes, num_queries, matcher, weight_dict, eos_coef, losses, data_file, evaluation=False):
        super().__init__()
        self.weight, self.evaluation = weight, evaluation
        self.num_classes, self.num_queries = num_classes, num_queries
        self.matcher, self.weight_dict = matcher, weight_dict
        self.eos_coef, self.losses, self.data_file = eos_coef, losses, data_file

    def loss_boxes(self, outputs, targets, indices, num_boxes):
        idx = _src_idx(indices, outputs["pred_boxes"].device)
        src = outputs["pred_boxes"][idx].float()
        tgt = torch.cat([t["boxes"][i.to(t["boxes"].device)] for t, (_, i) in zip(targets, indices)], dim=0)[:, 1:].float()
        if src.shape[0] == 0:
            z = src.sum() * 0
            return {"loss_bbox": z, "loss_giou": z}
        l1 = (src - tgt).abs().sum() / num_boxes
        giou = torch.diag(box_ops.generalized_box_iou(box_ops.box_cxcywh_to_xyxy(src), box_ops.box_cxcywh_to_xyxy(tgt)))
        return {"loss_bbox": l1, "loss_giou": (1 - giou).sum() / num_boxes}

    def _layers(self, outputs):
        main = {k: v for k, v in outputs.items() if k != "aux_outputs"}
        return [main] + list(outputs.get("aux_outputs", []))

    def forward(self, outputs, targets):
        layers = [self._select(o, targets) for o in self._layers(outputs)]
        all_idx = self.matcher.match_layers(layers, targets)
        num_boxes = float(sum(len(t["labels"]) for t in targets))     # local count, not all-reduced (criterion.py:182-183)
        losses = {}
        for li, (o, indices) in enumerate(zip(layers, all_idx)):
            ld = self.loss_labels(o, targets, indices, num_boxes, log=(li == 0))
            ld.update(self.loss_boxes(o, targets, indices, max(num_boxes, 1.0)))
            losses.update(ld if li == 0 else {k + "_%d" % (li - 1): v for k, v in ld.items()})
        self.last_indices = all_idx
        return losses

    def _select(self, o, targets):
        return o


class SetCriterionAVA(_SetCriterionBase):
    """AVA: 3-way actor CE (target 1 matched / 2 unmatched, weights [1,1,eos]) + weighted multi-label BCE (criterion.py:42-81)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        ew = torch.ones(3)
        ew[-1] = self.eos_coef
        self.register_buffer("empty_weight", ew)

    def loss_labels(self, outputs, targets, indices, num_boxes, log=True):
        lg, lb = outputs["pred_logits"].float(), outputs["pred_logits_b"].float()
        dev = lg.device
        idx = _src_idx(indices, dev)
        tcb = torch.full(lb.shape[:2], 2, dtype=torch.int64, device=dev)
        tcb[idx] = 1
        loss_ce_b = F.cross_entropy(lb.transpose(1, 2), tcb, self.empty_weight.to(dev))
        tco = torch.cat([t["labels"][J.to(t["labels"].device)] for t, (_, J) in zip(targets, indices)]).float()
        tc = torch.zeros_like(lg)
        tc[idx] = tco
        # BCE(sigmoid(x), t) = softplus(x) - t*x ; the reference clamps each log term at -100
        per = torch.minimum(F.softplus(-lg), lg.new_tensor(100.0)) * tc + torch.minimum(F.softplus(lg), lg.new_tensor(100.0)) * (1 - tc)
        if not self.evaluation:
            w = torch.ones(lg.shape[:2], dtype=lg.dtype, device=dev)
            w[idx] = self.weight
            per = per * w[:, :, None]
        losses = {"loss_ce": per.mean(), "loss_ce_b": loss_ce_b}
        if log:
            losses["class_error"] = 100 - accuracy_sigmoid(lg[idx], tco)[0]
        return losses


class SetCriterion(_SetCriterionBase):
    """JHMDB/UCF: (C+1)-way CE with no-object weight, 2-way visibility CE, key-frame query gather (criterion.py:237-262,378-396)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        ew = torch.ones(self.num_classes + 1)
        ew[-1] = self.eos_coef
        self.register_buffer("empty_weight", ew)

    def _select(self, o, targets):
        nq = self.num_queries
        dev = o["pred_logits"].device
        kf = torch.stack([nq * t["key_pos"].to(dev) + torch.arange(nq, device=dev) for t in targets])
        sel = {}
        for k, v in o.items():
            sel[k] = v.gather(1, kf[:, :, None].expand(-1, -1, v.shape[-1])) if k in ("pred_boxes", "pred_logits") else v
        return sel

    def loss_labels(self, outputs, targets, indices, num_boxes, log=True):
        lg, lb = outputs["pred_logits"].float(), outputs["pred_logits_b"].float()
        dev = lg.device
        idx = _src_idx(indices, dev)
        vis = torch.cat([t["vis"] for t in targets]).view(-1).to(dev)
        loss_ce_b = F.cross_entropy(lb, vis)
        tco = torch.cat([t["labels"][J.to(t["labels"].device)] for t, (_, J) in zip(targets, indices)]).to(dev)
        tc = torch.full(lg.shape[:2], self.num_classes, dtype=torch.int64, device=dev)
        tc[idx] = tco
        losses = {"loss_ce": F.cross_entropy(lg.transpose(1, 2), tc, self.empty_weight.to(dev)), "loss_ce_b": loss_ce_b}
        if log:
            losses["class_error"] = 100 - accuracy(lg[idx], tco)[0]
        return losses


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
