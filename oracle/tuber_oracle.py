"""CPU ORACLE for the TubeR hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch *functional* restatement, in stock PyTorch CPU ops
(fp32, or fp64 when the state dict is cast), of the reference's forward /
criterion / post-processing path.  It is a checker: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped model (``tubelet_transformer_amd``) never does and fails loudly when
its HIP library is missing.

Pinning: the reference holds no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself,
imported in the build container by ``oracle/gen_golden.py`` (<= 1e-5 abs), and the
resulting vectors are committed under ``tests/golden/``.

Every function takes a flat ``state`` dict (the model ``state_dict`` with the
reference's key names) and cites the reference lines it follows.
Third-party arithmetic: ``scipy.optimize.linear_sum_assignment`` (SciPy 1.15.3 here;
the reference does not pin a version -- call sites ``models/detr/matcher.py:80``,
``models/detr/matcher_ucf.py:82``).
"""
import math

import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

BN_EPS = 1e-3       # models/backbones/ir_CSN_152.py:15
BN_MOMENTUM = 0.1   # models/backbones/ir_CSN_152.py:16
CSN_BLOCKS = {"CSN-152": [3, 8, 36, 3], "CSN-50": [3, 4, 6, 3],   # ir_CSN_152.py:204 / ir_CSN_50.py:204
              "CSN-TEST": [2, 2, 2, 2]}   # shallow test-only body (same block code; keeps bf16 parity tests well-conditioned)


# --------------------------------------------------------------------------
# backbone
# --------------------------------------------------------------------------
def batch_norm(state, p, x, train):
    """nn.BatchNorm3d(eps=1e-3, momentum=0.1) -- ir_CSN_152.py:46,56,64,119,154.
    In train mode the running buffers in ``state`` are updated in place like the module does."""
    if train and (p + ".num_batches_tracked") in state:
        state[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, state[p + ".running_mean"], state[p + ".running_var"], state[p + ".weight"],
                        state[p + ".bias"], train, BN_MOMENTUM, BN_EPS)


def bottleneck(state, p, x, stride, tstride, has_ds, train):
    """ResNeXtBottleneck.forward -- ir_CSN_152.py:70-90: pw -> BN -> ReLU -> depthwise 3x3x3 (stride here)
    -> BN -> ReLU -> pw -> BN -> (+ shortcut) -> ReLU."""
    planes = state[p + ".conv1.weight"].shape[0]
    out = F.conv3d(x, state[p + ".conv1.weight"])
    out = F.relu(batch_norm(state, p + ".bn1", out, train))
    out = F.conv3d(out, state[p + ".conv3.weight"], stride=(tstride, stride, stride), padding=1, groups=planes)
    out = F.relu(batch_norm(state, p + ".bn3", out, train))
    out = F.conv3d(out, state[p + ".conv4.weight"])
    out = batch_norm(state, p + ".bn4", out, train)
    res = x
    if has_ds:  # ir_CSN_152.py:155-165: strided 1x1x1 projection + BN on every stage's first block
        res = F.conv3d(x, state[p + ".down_sample.0.weight"], stride=(tstride, stride, stride))
        res = batch_norm(state, p + ".down_sample.1", res, train)
    return F.relu(out + res)


def csn_body(state, p, x, backbone_name, last_stride, train):
    """ResNeXt.forward -- ir_CSN_152.py:172-186 (stem :109-122, stages :124-135)."""
    x = F.conv3d(x, state[p + ".conv1.weight"], stride=(1, 2, 2), padding=(1, 3, 3))
    x = F.relu(batch_norm(state, p + ".bn1", x, train))
    x = F.max_pool3d(x, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
    strides = [(1, 1), (2, 2), (2, 2), (2 if last_stride else 1, 2)]  # (spatial, temporal)
    for li, nblocks in enumerate(CSN_BLOCKS[backbone_name]):
        for bi in range(nblocks):
            s, ts = strides[li] if bi == 0 else (1, 1)
            x = bottleneck(state, "%s.layer%d.%d" % (p, li + 1, bi), x, s, ts, bi == 0, train)
    return x


def layer_norm(state, p, x):
    return F.layer_norm(x, (x.shape[-1],), state[p + ".weight"], state[p + ".bias"], 1e-5)


def linear(state, p, x):
    return F.linear(x, state[p + ".weight"], state[p + ".bias"])


def mha(state, p, q, k, v, nhead, key_padding_mask=None):
    """Multi-head attention with packed in-projection, as nn.MultiheadAttention and the
    reference's hand-rolled copy (transformer_layers.py:306-366,156-167) compute it:
    rows [q;k;v] of in_proj_weight, q scaled by 1/sqrt(d_h) AFTER projection, padded keys
    -> -inf, softmax, (dropout omitted: parity runs use p=0 / eval), AV, out_proj.
    q: (Lq,B,E); k,v: (Lk,B,E); key_padding_mask: (B,Lk) bool, True = ignore."""
    lq, b, e = q.shape
    lk = k.shape[0]
    hd = e // nhead
    w, bias = state[p + ".in_proj_weight"], state[p + ".in_proj_bias"]
    qp = F.linear(q, w[:e], bias[:e]) * (float(hd) ** -0.5)
    kp = F.linear(k, w[e:2 * e], bias[e:2 * e])
    vp = F.linear(v, w[2 * e:], bias[2 * e:])
    qp = qp.reshape(lq, b * nhead, hd).transpose(0, 1)
    kp = kp.reshape(lk, b * nhead, hd).transpose(0, 1)
    vp = vp.reshape(lk, b * nhead, hd).transpose(0, 1)
    s = torch.bmm(qp, kp.transpose(1, 2))
    if key_padding_mask is not None:
        s = s.view(b, nhead, lq, lk).masked_fill(key_padding_mask[:, None, None, :], float("-inf")).view(b * nhead, lq, lk)
    a = F.softmax(s, dim=-1)
    o = torch.bmm(a, vp).transpose(0, 1).reshape(lq, b, e)
    return linear(state, p + ".out_proj", o)


def lstr_decode_pool(state, p, xs, nhead=8):
    """TEMPORAL_DS_STRATEGY 'decode' -- backbone_builder.py:74-78 with
    LSTRTransformerDecoder[Layer] (transformer_layers.py:380-448): one learned query per pixel
    attends over the t temporal slots (d=2048)."""
    bs, ch, t, a, b = xs.shape
    mem = xs.reshape(bs, ch, t, a * b).permute(2, 0, 3, 1).reshape(t, bs * a * b, ch)
    tgt = state[p + ".query_pool.weight"].unsqueeze(1).repeat(1, bs * a * b, 1)
    L = p + ".pool_decoder.layers.0"
    tgt = layer_norm(state, L + ".norm1", tgt + mha(state, L + ".self_attn", tgt, tgt, tgt, nhead))
    tgt = layer_norm(state, L + ".norm2", tgt + mha(state, L + ".multihead_attn", tgt, mem, mem, nhead))
    ff = linear(state, L + ".linear2", F.relu(linear(state, L + ".linear1", tgt)))
    tgt = layer_norm(state, L + ".norm3", tgt + ff)
    tgt = layer_norm(state, p + ".pool_decoder.norm", tgt)
    return tgt.view(1, bs, a * b, ch).permute(1, 3, 0, 2).reshape(bs, ch, 1, a, b)


def position_embedding_sine_3d(mask, hidden_dim=256, temperature=10000.0):
    """PositionEmbeddingSine_3D(hidden_dim, normalize=True) -- position_encoding.py:32-72.
    mask (B,T,H,W) bool -> (B,hidden_dim,T,H,W); channels = [t: d/4 | y: 3d/8 | x: 3d/8]."""
    nt, ns = hidden_dim / 8 * 2, hidden_dim / 8 * 3
    not_mask = ~mask
    scale, eps = 2 * math.pi, 1e-6
    t_e = not_mask.cumsum(1, dtype=torch.float32)
    y_e = not_mask.cumsum(2, dtype=torch.float32)
    x_e = not_mask.cumsum(3, dtype=torch.float32)
    t_e = t_e / (t_e[:, -1:, :, :] + eps) * scale
    y_e = y_e / (y_e[:, :, -1:, :] + eps) * scale
    x_e = x_e / (x_e[:, :, :, -1:] + eps) * scale
    dt = torch.arange(nt, dtype=torch.float32)
    dt = temperature ** (2 * (dt // 2) / nt)
    dsp = torch.arange(ns, dtype=torch.float32)
    dsp = temperature ** (2 * (dsp // 2) / ns)

    def interleave(e, d):
        pp = e[..., None] / d
        return torch.stack((pp[..., 0::2].sin(), pp[..., 1::2].cos()), dim=5).flatten(4)

    pos = torch.cat((interleave(t_e, dt), interleave(y_e, dsp), interleave(x_e, dsp)), dim=4)
    return pos.permute(0, 4, 1, 2, 3)


def backbone_forward(state, cfg, clips, mask, train):
    """Backbone.forward -- backbone_builder.py:59-90.  Returns (xs, mask', pos, xt)."""
    M = cfg.CONFIG.MODEL
    xs = csn_body(state, "backbone.body", clips, M.BACKBONE_NAME, M.LAST_STRIDE, train)
    xt = xs
    t = xs.shape[2]
    if M.SINGLE_FRAME:
        if M.TEMPORAL_DS_STRATEGY == "avg":
            xs = F.avg_pool3d(xs, (M.TEMP_LEN // M.DS_RATE, 1, 1))
        elif M.TEMPORAL_DS_STRATEGY == "max":
            xs = F.max_pool3d(xs, (M.TEMP_LEN // M.DS_RATE, 1, 1))
        elif M.TEMPORAL_DS_STRATEGY == "decode":
            xs = lstr_decode_pool(state, "backbone", xs)
        else:  # any other string (JHMDB ships 'decoder'): mid-frame slice, :79-80
            xs = xs[:, :, t // 2: t // 2 + 1]
    m = F.interpolate(mask[None].float(), size=xs.shape[-2:]).to(torch.bool)[0]
    m = m.unsqueeze(1).repeat(1, xs.shape[2], 1, 1)
    pos = position_embedding_sine_3d(m, M.D_MODEL).to(xs.dtype)
    return xs, m, pos, xt


# --------------------------------------------------------------------------
# DETR encoder / decoder (models/transformer/transformer.py)
# --------------------------------------------------------------------------
def detr_transformer(state, p, src, mask, query_embed, pos, nhead, n_enc, n_dec):
    """Transformer.forward -- transformer.py:49-64; post-norm layers :153-168 (enc), :218-249 (dec);
    decoder.norm applied to every intermediate output :116-126.  Returns hs (n_dec,B,Q,E)."""
    bs = src.shape[0]
    x = src.flatten(2).permute(2, 0, 1)
    pe = pos.flatten(2).permute(2, 0, 1)
    qpos = query_embed.unsqueeze(1).repeat(1, bs, 1)
    kpm = mask.flatten(1)
    for i in range(n_enc):
        L = "%s.encoder.layers.%d" % (p, i)
        qk = x + pe
        x = layer_norm(state, L + ".norm1", x + mha(state, L + ".self_attn", qk, qk, x, nhead, kpm))
        ff = linear(state, L + ".linear2", F.relu(linear(state, L + ".linear1", x)))
        x = layer_norm(state, L + ".norm2", x + ff)
    memory = x
    tgt = torch.zeros_like(qpos)
    outs = []
    for i in range(n_dec):
        L = "%s.decoder.layers.%d" % (p, i)
        qk = tgt + qpos
        tgt = layer_norm(state, L + ".norm1", tgt + mha(state, L + ".self_attn", qk, qk, tgt, nhead))
        ca = mha(state, L + ".multihead_attn", tgt + qpos, memory + pe, memory, nhead, kpm)
        tgt = layer_norm(state, L + ".norm2", tgt + ca)
        ff = linear(state, L + ".linear2", F.relu(linear(state, L + ".linear1", tgt)))
        tgt = layer_norm(state, L + ".norm3", tgt + ff)
        outs.append(layer_norm(state, p + ".decoder.norm", tgt))
    return torch.stack(outs).transpose(1, 2)


def class_branch_encoder(state, p, src, shape5, nhead=8):
    """Factorised t/s encoder layer -- transformer_layers.py:71-97.  NB the naming is swapped in the
    reference: ``self_attn_t`` attends over the h*w spatial tokens, ``self_attn_s`` over the t slots."""
    _, ch, t, h, w = shape5
    bs = src.shape[1]
    src_t = src.view(t, h * w, bs, ch).permute(1, 0, 2, 3).reshape(h * w, t * bs, ch)
    src_t = layer_norm(state, p + ".norm1_t", src_t + mha(state, p + ".self_attn_t", src_t, src_t, src_t, nhead))
    src_t = src_t.view(h * w, t, bs, ch).permute(1, 0, 2, 3).reshape(t * h * w, bs, ch)
    src_s = src.reshape(t, h * w * bs, ch)
    src_s = layer_norm(state, p + ".norm1_s", src_s + mha(state, p + ".self_attn_s", src_s, src_s, src_s, nhead))
    src_s = src_s.view(t * h * w, bs, ch)
    cat = torch.cat((src_t, src_s), dim=-1)
    ff = linear(state, p + ".linear2", F.relu(linear(state, p + ".linear1", cat)))
    return layer_norm(state, p + ".norm2", src + ff)


def nested_from_list(clips):
    """nested_tensor_from_tensor_list -- utils/misc.py:367-402 (4-D clips branch)."""
    if isinstance(clips, torch.Tensor):
        clips = list(clips)
    mx = [max(c.shape[d] for c in clips) for d in range(4)]
    out = torch.zeros([len(clips)] + mx, dtype=clips[0].dtype)
    mask = torch.ones((len(clips), mx[2], mx[3]), dtype=torch.bool)
    for i, c in enumerate(clips):
        out[i, : c.shape[0], : c.shape[1], : c.shape[2], : c.shape[3]] = c
        mask[i, : c.shape[2], : c.shape[3]] = False
    return out, mask


def tuber_forward(state, cfg, clips, mask=None, train=False):
    """DETR.forward -- models/tuber_ava.py:97-148 (dropout disabled: parity runs use eval or p=0)."""
    M = cfg.CONFIG.MODEL
    ava = cfg.CONFIG.DATA.DATASET_NAME == "ava"
    if mask is None:
        clips, mask = nested_from_list(clips)
    xs, m, pos, xt = backbone_forward(state, cfg, clips, mask, train)
    src = F.conv3d(xs, state["input_proj.weight"], state["input_proj.bias"])
    hs = detr_transformer(state, "transformer", src, m, state["query_embed.weight"], pos,
                          M.NHEAD, M.ENC_LAYERS, M.DEC_LAYERS)
    lay_n, bs, nb, dim = hs.shape
    if ava:
        logits_b = linear(state, "class_embed_b", hs)
    else:  # tuber_ava.py:124-125 (literal 6)
        pooled = xt.mean(dim=(2, 3, 4))
        logits_b = linear(state, "class_embed_b", pooled).unsqueeze(0).repeat(6, 1, 1)
    src_c = F.conv3d(xt, state["class_proj.weight"], state["class_proj.bias"])
    flat = src_c.view(1, bs, dim, -1).repeat(lay_n, 1, 1, 1).view(lay_n * bs, dim, -1).permute(2, 0, 1).contiguous()
    flat = class_branch_encoder(state, "encoder.layers.0", flat, src_c.shape)
    hq = hs.reshape(lay_n * bs, nb, dim).permute(1, 0, 2)
    qc = mha(state, "cross_attn", hq, flat, flat, 8)
    qc = qc.permute(1, 0, 2).reshape(lay_n, bs, nb, dim)
    logits = linear(state, "class_fc", qc)
    x = hs
    for i in range(3):  # MLP(256,256,4,3) -- criterion.py:485-497
        x = linear(state, "bbox_embed.layers.%d" % i, x)
        if i < 2:
            x = F.relu(x)
    boxes = x.sigmoid()
    out = {"pred_logits": logits[-1], "pred_boxes": boxes[-1], "pred_logits_b": logits_b[-1]}
    if cfg.CONFIG.TRAIN.AUX_LOSS:
        out["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b, "pred_logits_b": c}
                              for a, b, c in zip(logits[:-1], boxes[:-1], logits_b[:-1])]
    return out


# --------------------------------------------------------------------------
# boxes, matcher, criterion, post-processing
# --------------------------------------------------------------------------
def cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def pairwise_giou(b1, b2):
    """generalized_box_iou -- utils/box_ops.py:41-65 (with its degenerate-box asserts :55-56)."""
    assert (b1[:, 2:] >= b1[:, :2]).all()
    assert (b2[:, 2:] >= b2[:, :2]).all()
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2 - inter
    iou = inter / union
    whc = (torch.max(b1[:, None, 2:], b2[:, 2:]) - torch.min(b1[:, None, :2], b2[:, :2])).clamp(min=0)
    area = whc[..., 0] * whc[..., 1]
    return iou - (area - union) / area


@torch.no_grad()
def hungarian_match(cfg, outputs, targets):
    """HungarianMatcher.forward -- matcher.py:37-81 (AVA) / matcher_ucf.py:37-88 (JHMDB)."""
    Mc = cfg.CONFIG.MATCHER
    ava = cfg.CONFIG.DATA.DATASET_NAME == "ava"
    bs, nq = (outputs["pred_logits_b"] if ava else outputs["pred_logits"]).shape[:2]
    ob = outputs["pred_boxes"].flatten(0, 1)
    tb = torch.cat([t["boxes"] for t in targets])[:, 1:]
    c_bbox = torch.cdist(ob, tb, p=1)
    c_giou = -pairwise_giou(cxcywh_to_xyxy(ob), cxcywh_to_xyxy(tb))
    if ava:
        prob = outputs["pred_logits_b"].flatten(0, 1).softmax(-1)
        c_cls = -prob[:, 1:2].repeat(1, len(tb))
    else:
        ids = torch.cat([t["labels"] for t in targets])
        c_cls = -outputs["pred_logits"].flatten(0, 1).softmax(-1)[:, ids]
    C = (Mc.COST_BBOX * c_bbox + Mc.COST_CLASS * c_cls + Mc.COST_GIOU * c_giou).view(bs, nq, -1).cpu()
    sizes = [len(t["boxes"]) for t in targets]
    idx = [linear_sum_assignment(c[i]) for i, c in enumerate(C.split(sizes, -1))]
    return [(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)) for i, j in idx], C


def _src_idx(indices):
    b = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
    return b, torch.cat([s for s, _ in indices])


def _losses_ava(cfg, out, targets, indices, num_boxes):
    """SetCriterionAVA.loss_labels / loss_boxes -- criterion.py:42-81,97-117."""
    Lc = cfg.CONFIG.LOSS_COFS
    idx = _src_idx(indices)
    lb = out["pred_logits_b"]
    tcb = torch.full(lb.shape[:2], 2, dtype=torch.int64)
    tcb[idx] = 1
    w3 = torch.ones(3, dtype=lb.dtype)
    w3[-1] = Lc.EOS_COF
    loss_ce_b = F.cross_entropy(lb.transpose(1, 2), tcb, w3)
    lg = out["pred_logits"]
    tco = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
    tc = torch.zeros_like(lg)
    tc[idx] = tco.to(lg.dtype)
    wts = torch.ones(lg.shape[:2], dtype=lg.dtype)
    wts[idx] = Lc.WEIGHT
    if cfg.CONFIG.EVAL_ONLY:
        loss_ce = F.binary_cross_entropy(lg.sigmoid(), tc)
    else:
        loss_ce = F.binary_cross_entropy(lg.sigmoid(), tc, weight=wts[:, :, None])
    return {"loss_ce": loss_ce, "loss_ce_b": loss_ce_b, **_losses_boxes(out, targets, indices, num_boxes)}


def _losses_boxes(out, targets, indices, num_boxes):
    idx = _src_idx(indices)
    sb = out["pred_boxes"][idx]
    tb = torch.cat([t["boxes"][i] for t, (_, i) in zip(targets, indices)], dim=0)[:, 1:].to(sb.dtype)
    l1 = (sb - tb).abs().sum() / num_boxes
    giou = (1 - torch.diag(pairwise_giou(cxcywh_to_xyxy(sb), cxcywh_to_xyxy(tb)))).sum() / num_boxes
    return {"loss_bbox": l1, "loss_giou": giou}


def _losses_jhmdb(cfg, out, targets, indices, num_boxes):
    """SetCriterion.loss_labels / loss_boxes -- criterion.py:237-262,280-318."""
    nc = cfg.CONFIG.DATA.NUM_CLASSES
    idx = _src_idx(indices)
    vis = torch.cat([t["vis"] for t in targets]).view(-1)
    loss_ce_b = F.cross_entropy(out["pred_logits_b"], vis)
    lg = out["pred_logits"]
    tco = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
    tc = torch.full(lg.shape[:2], nc, dtype=torch.int64)
    tc[idx] = tco
    ew = torch.ones(nc + 1, dtype=lg.dtype)
    ew[-1] = cfg.CONFIG.LOSS_COFS.EOS_COF
    loss_ce = F.cross_entropy(lg.transpose(1, 2), tc, ew)
    return {"loss_ce": loss_ce, "loss_ce_b": loss_ce_b, **_losses_boxes(out, targets, indices, num_boxes)}


def set_criterion(cfg, outputs, targets):
    """SetCriterionAVA.forward -- criterion.py:169-206 / SetCriterion.forward -- :366-410.
    Returns (loss dict without class_error, list of matcher indices per decoder layer [last, aux0..])."""
    ava = cfg.CONFIG.DATA.DATASET_NAME == "ava"
    nq = cfg.CONFIG.MODEL.QUERY_NUM
    fn = _losses_ava if ava else _losses_jhmdb

    def keyframes(o):  # criterion.py:378-380: JHMDB gathers the key-frame queries key_pos*nq + j
        if ava:
            return o
        kf = torch.stack([nq * t["key_pos"].cpu() + torch.arange(nq) for t in targets])
        sel = {}
        for k, v in o.items():
            sel[k] = v.gather(1, kf[:, :, None].repeat(1, 1, v.shape[-1])) if k in ("pred_boxes", "pred_logits") else v
        return sel

    num_boxes = float(sum(len(t["labels"]) for t in targets))
    layers = [{k: v for k, v in outputs.items() if k != "aux_outputs"}] + list(outputs.get("aux_outputs", []))
    losses, all_idx = {}, []
    for li, o in enumerate(layers):
        o = keyframes(o)
        indices, _ = hungarian_match(cfg, o, targets)
        all_idx.append(indices)
        ld = fn(cfg, o, targets, indices, num_boxes)
        losses.update(ld if li == 0 else {k + "_%d" % (li - 1): v for k, v in ld.items()})
    return losses, all_idx


def weight_dict(cfg):
    """models/tuber_ava.py:185-196."""
    Lc = cfg.CONFIG.LOSS_COFS
    wd = {"loss_ce": Lc.DICE_COF, "loss_bbox": Lc.BBOX_COF, "loss_giou": Lc.GIOU_COF, "loss_ce_b": 1}
    if cfg.CONFIG.TRAIN.AUX_LOSS:
        for i in range(cfg.CONFIG.MODEL.DEC_LAYERS - 1):
            wd.update({k + "_%d" % i: v for k, v in list(wd.items())[:4]})
    return wd


def total_loss(cfg, loss_dict):
    """utils/video_action_recognition.py:144-148."""
    wd = weight_dict(cfg)
    return sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)


@torch.no_grad()
def post_process(cfg, outputs, target_sizes):
    """PostProcessAVA.forward -- criterion.py:447-482 / PostProcess.forward -- :413-445."""
    lb, lg, bx = outputs["pred_logits_b"], outputs["pred_logits"], outputs["pred_boxes"]
    boxes = cxcywh_to_xyxy(bx)
    h, w = target_sizes.unbind(1)
    boxes = boxes * torch.stack([w, h, w, h], dim=1)[:, None, :]
    if cfg.CONFIG.DATA.DATASET_NAME == "ava":
        pb = lb.softmax(-1)[:, :, 1:2]
        prob = lg.sigmoid() * ((pb > 0.8).float() * pb)
        return prob.numpy(), boxes.numpy(), pb.numpy()
    return F.softmax(lg, -1).numpy(), boxes.numpy(), lb.softmax(-1).numpy()[..., 1:]
