"""TubeR model API on the MI355X HIP path.

Mirrors ``models/tuber_ava.py`` (``DETR`` :22-157, ``build_model`` :160-221), ``models/backbone_builder.py``
(``Backbone`` :26-90), ``models/transformer/transformer.py`` and ``transformer_layers.py``: same module tree,
attribute names and ``state_dict`` keys, so released checkpoints load and ``train_tuber_*.py`` / ``eval_tuber_*.py``
style drivers work unchanged.  The torch.nn children hold parameters only; ``DETR.forward`` runs the whole
network through libtuber_hip.so (NDHWC / token-major bf16 activations, fp32 master weights) and raises if the
library or a GPU is missing -- there is no eager fallback.

Row orders used internally (all row-wise ops are order-agnostic; attention gets explicit strides):
    backbone features   (b, t, hw)      encoder memory   (b, hw)        decoder / hs   (layer, b, query)
    class branch        (layer*B + b, t, hw)   -- the reference's permute/contiguous copies
    (tuber_ava.py:133-139, transformer_layers.py:77-91) are never materialised.
"""
import copy

import torch
from torch import nn

from . import ab, lib
from . import tape as T
from .backbone import build_CSN, CSNRunner
from .engine import ParamStore
from .misc import NestedTensor, nested_tensor_from_tensor_list

BF = torch.bfloat16


# ------------------------------------------------------------------------------------------------
# parameter containers with the reference's names
# ------------------------------------------------------------------------------------------------
def _clones(m, n):
    return nn.ModuleList([copy.deepcopy(m) for _ in range(n)])


class TransformerEncoderLayer(nn.Module):      # transformer.py:131-189
    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)


class TransformerDecoderLayer(nn.Module):      # transformer.py:192-285
    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)


class _Stack(nn.Module):
    def __init__(self, layer, n, norm=None):
        super().__init__()
        self.layers = _clones(layer, n)
        self.num_layers = n
        self.norm = norm


class Transformer(nn.Module):                  # transformer.py:15-64
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, normalize_before=False):
        super().__init__()
        if normalize_before:
            raise NotImplementedError("NORMALIZE_BEFORE: True is broken in the reference (transformer.py:81,170-182); "
                                      "only the post-norm path is implemented")
        self.encoder = _Stack(TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout), num_encoder_layers)
        self.decoder = _Stack(TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout), num_decoder_layers,
                              nn.LayerNorm(d_model))
        for p in self.parameters():            # transformer.py:44-47
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.d_model, self.nhead = d_model, nhead


def build_transformer(cfg):                    # transformer.py:302-314
    M = cfg.CONFIG.MODEL
    return Transformer(M.D_MODEL, M.NHEAD, M.ENC_LAYERS, M.DEC_LAYERS, M.DIM_FEEDFORWARD, M.DROPOUT, M.NORMALIZE_BEFORE)


class ClassEncoderLayer(nn.Module):            # transformer_layers.py:46-69 (factorised t/s layer)
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn_t = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.self_attn_s = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model * 2, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1_t = nn.LayerNorm(d_model)
        self.norm1_s = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)


class LSTRDecoderLayer(nn.Module):             # transformer_layers.py:403-448 (d=2048 pool decoder)
    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)


class MLP(nn.Module):                          # criterion.py:485-497
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))


class Backbone(nn.Module):                     # backbone_builder.py:26-57
    def __init__(self, train_backbone, num_channels, cfg):
        super().__init__()
        self.body = build_CSN(cfg)
        if not train_backbone:
            for p in self.body.parameters():
                p.requires_grad_(False)
        M = cfg.CONFIG.MODEL
        self.ds = M.SINGLE_FRAME
        self.pool_len = M.TEMP_LEN // M.DS_RATE
        if M.SINGLE_FRAME and M.TEMPORAL_DS_STRATEGY == "decode":
            self.query_pool = nn.Embedding(1, 2048)
            self.pool_decoder = _Stack(LSTRDecoderLayer(2048, 8, 2048, 0.1), 1, nn.LayerNorm(2048))
        self.num_channels = num_channels
        self.backbone_name = M.BACKBONE_NAME
        self.temporal_ds_strategy = M.TEMPORAL_DS_STRATEGY


def build_backbone(cfg):                       # backbone_builder.py:109-113
    return Backbone(cfg.CONFIG.TRAIN.LR_BACKBONE > 0, cfg.CONFIG.MODEL.DIM_FEEDFORWARD, cfg)


# ------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------
class DETR(nn.Module):
    """TubeR detector (tuber_ava.py:22-157).  ``forward(samples)`` -> dict with ``pred_logits`` [B,Q,C],
    ``pred_boxes`` [B,Q,4] (cxcywh in (0,1)), ``pred_logits_b`` and ``aux_outputs`` -- fp32 tensors on the GPU."""

    def __init__(self, backbone, transformer, num_classes, num_queries, hidden_dim, temporal_length, aux_loss=False,
                 generate_lfb=False, backbone_name="CSN-152", ds_rate=1, last_stride=True, dataset_mode="ava"):
        super().__init__()
        self.temporal_length = temporal_length
        self.num_queries = num_queries
        self.transformer = transformer
        self.avg = nn.AvgPool3d(kernel_size=(temporal_length, 1, 1))
        self.dataset_mode = dataset_mode
        if dataset_mode != "ava":
            self.avg_s = nn.AdaptiveAvgPool3d((1, 1, 1))
            self.query_embed = nn.Embedding(num_queries * temporal_length, hidden_dim)
        else:
            self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.input_proj = nn.Conv3d(backbone.num_channels, hidden_dim, kernel_size=1)
        self.class_proj = nn.Conv3d(backbone.num_channels, hidden_dim, kernel_size=1)
        self.encoder = _Stack(ClassEncoderLayer(hidden_dim, 8, 2048, 0.1), 1)
        self.cross_attn = nn.MultiheadAttention(256, num_heads=8, dropout=0.1)
        self.class_embed_b = nn.Linear(hidden_dim, 3) if dataset_mode == "ava" else nn.Linear(2048, 2)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.class_fc = nn.Linear(hidden_dim, num_classes if dataset_mode == "ava" else num_classes + 1)
        self.dropout = nn.Dropout(0.5)
        self.backbone = backbone
        self.aux_loss = aux_loss
        self.hidden_dim = hidden_dim
        self.generate_lfb = generate_lfb
        self.last_stride = last_stride
        self._store = None
        self._runner = None
        self._anchor = None

    # -- engine ----------------------------------------------------------------------------------
    def engine(self):
        """(ParamStore, CSNRunner) for the device the parameters live on; (re)built when parameters moved."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("TubeR MI355X path: parameters are on %s; move the model to a ROCm GPU (model.cuda()). "
                               "There is no CPU fallback." % dev)
        if self._store is None or not self._store.valid() or self._store.device != dev:
            lib.load()
            self._store = ParamStore(self, dev)
            self._runner = CSNRunner(self.backbone.body, "backbone.body.", self._store)
            self._anchor = torch.zeros((), device=dev, requires_grad=True)
            self._store.anchor = self._anchor
        return self._store, self._runner

    def freeze_params(self):                   # tuber_ava.py:83-95
        for mod in (self.backbone, self.transformer, self.query_embed, self.bbox_embed, self.input_proj, self.class_embed_b):
            for p in mod.parameters():
                p.requires_grad = False

    # -- helpers -----------------------------------------------------------------------------------
    def _mha_self(self, tp, x, pos, prefix, B, L, kpm, p):
        """self-attention with q = k = x + pos, v = x (rows (b, l)); returns out_proj(attn).  p = attention-weight dropout.
        The packed in-projection is ONE GEMM: its q / k rows see x + pos (added while the operand is staged), its v rows x."""
        E = self.hidden_dim
        qkv = T.in_proj(tp, x, pos, prefix + ".in_proj_weight", prefix + ".in_proj_bias", (0, 3 * E), 2 * E)
        a = T.attention(tp, ((0, 0), (0, E), (0, 2 * E)), (B, 8, L, L, (1, L, 0, 1), (1, L, 0, 1)), kpm, p, qkv)
        return T.linear(tp, a, prefix + ".out_proj.weight", prefix + ".out_proj.bias")

    def _ffn(self, tp, x, prefix, p):
        """linear2(Dropout(ReLU(linear1(x)))); the Dropout that follows linear2 is fused into the caller's LayerNorm."""
        h = T.linear(tp, x, prefix + ".linear1.weight", prefix + ".linear1.bias", relu=True, drop=p)
        return T.linear(tp, h, prefix + ".linear2.weight", prefix + ".linear2.bias")

    def _decoder_f32(self, tp, st, memory, pos, kpm, hs, B, Q, Lm, lay_n, E, H):
        """Eval precision mode: TransformerDecoder.forward (models/transformer/transformer.py:99-128,218-249) in fp32 on the master
        parameters -- packed in-projections with the positional embeddings folded in, fp32 attention cores, post-norm LayerNorms on the
        fp32 residual stream, FFN -- through tuber_linear_f32 / tuber_attention_f32 / tuber_layernorm_fwd_f32.  The encoder memory enters
        as its fp32 LayerNorm twin (tape.f32).  Writes hs (bf16 rows (layer, b, q): the class branch's query operand) and returns the
        fp32 decoder output the box / actor heads read."""
        dev, F32 = st.device, torch.float32
        fp = lambda name: st.flat.data_ptr() + 4 * st.offsets[name]
        R, FF = B * Q, self.transformer.decoder.layers[0].linear1.out_features
        scale = float(E // H) ** -0.5
        tw = tp.f32.get(id(memory))
        if tw is not None:
            mem32 = tw[1]
        else:
            mem32 = torch.empty(memory.shape, dtype=F32, device=dev)
            lib.call("tuber_cast_bf16_f32_scale", memory, mem32, memory.numel(), 1.0)
        pos32 = torch.empty(pos.shape, dtype=F32, device=dev)
        lib.call("tuber_cast_bf16_f32_scale", pos, pos32, pos.numel(), 1.0)
        o = st.offsets["query_embed.weight"]
        qpos32 = st.flat[o:o + Q * E].view(Q, E).repeat(B, 1)                       # rows (b, q): the fp32 query embeddings themselves
        tgt = torch.zeros(R, E, dtype=F32, device=dev)
        hs32 = torch.empty(lay_n * R, E, dtype=F32, device=dev)
        scratch = torch.empty(R, E, dtype=torch.bfloat16, device=dev)               # the bf16 twin of a LayerNorm output nobody reads here

        def lin(x, ldx, add, addc, w, b, y, ldy, M, N, K, act=0):
            lib.call("tuber_linear_f32", x, ldx, add, E if add is not None else 0, addc, w, K, b, y, ldy, M, N, K, act)

        # the memory-side projections [(memory + pos) W_k | memory W_v] of ALL layers do not depend on the decoder state: one launch over the
        # layers' weight sets (the parameters of consecutive layers sit at a constant stride in the flat buffer)
        P0, P1 = "transformer.decoder.layers.0.multihead_attn", "transformer.decoder.layers.%d.multihead_attn" % min(1, lay_n - 1)
        wz = st.offsets[P1 + ".in_proj_weight"] - st.offsets[P0 + ".in_proj_weight"]
        bz = st.offsets[P1 + ".in_proj_bias"] - st.offsets[P0 + ".in_proj_bias"]
        uniform = wz % 4 == 0 and all(st.offsets["transformer.decoder.layers.%d.multihead_attn.in_proj_weight" % i] == st.offsets[P0 + ".in_proj_weight"] + i * wz and
                      st.offsets["transformer.decoder.layers.%d.multihead_attn.in_proj_bias" % i] == st.offsets[P0 + ".in_proj_bias"] + i * bz for i in range(lay_n))
        kvs = torch.empty(lay_n, B * Lm, 2 * E, dtype=F32, device=dev)
        if uniform:
            lib.call("tuber_linear_f32_batched", mem32, E, pos32, E, E, fp(P0 + ".in_proj_weight") + 4 * E * E, E, fp(P0 + ".in_proj_bias") + 4 * E, kvs, 2 * E,
                     B * Lm, 2 * E, E, 0, lay_n, wz, bz, B * Lm * 2 * E)
        else:
            for i in range(lay_n):
                P = "transformer.decoder.layers.%d.multihead_attn" % i
                lin(mem32, E, pos32, E, fp(P + ".in_proj_weight") + 4 * E * E, fp(P + ".in_proj_bias") + 4 * E, kvs[i], 2 * E, B * Lm, 2 * E, E)

        def norm(x32, res32, prefix, y=None, y32=None):
            out32 = torch.empty(R, E, dtype=F32, device=dev) if y32 is None else y32
            lib.call("tuber_layernorm_fwd_f32", None, x32, None, res32, fp(prefix + ".weight"), fp(prefix + ".bias"), scratch if y is None else y, E, out32, R, E, 1e-5)
            return out32
        for i in range(lay_n):
            L = "transformer.decoder.layers.%d" % i
            S, P = L + ".self_attn", L + ".multihead_attn"
            qkv = torch.empty(R, 3 * E, dtype=F32, device=dev)
            lin(tgt, E, qpos32, 2 * E, fp(S + ".in_proj_weight"), fp(S + ".in_proj_bias"), qkv, 3 * E, R, 3 * E, E)        # q | k see tgt + query_pos, v sees tgt
            a = torch.empty(R, E, dtype=F32, device=dev)
            lib.call("tuber_attention_f32", qkv, 3 * E, qkv.data_ptr() + 4 * E, 3 * E, qkv.data_ptr() + 8 * E, 3 * E, a, E, None, B, H, Q, Q, scale)
            ao = torch.empty(R, E, dtype=F32, device=dev)
            lin(a, E, None, 0, fp(S + ".out_proj.weight"), fp(S + ".out_proj.bias"), ao, E, R, E, E)
            tgt = norm(ao, tgt, L + ".norm1")
            q = torch.empty(R, E, dtype=F32, device=dev)
            lin(tgt, E, qpos32, E, fp(P + ".in_proj_weight"), fp(P + ".in_proj_bias"), q, E, R, E, E)                      # (tgt + query_pos) W_q
            kv = kvs[i]
            lib.call("tuber_attention_f32", q, E, kv, 2 * E, kv.data_ptr() + 4 * E, 2 * E, a, E, kpm, B, H, Q, Lm, scale)
            lin(a, E, None, 0, fp(P + ".out_proj.weight"), fp(P + ".out_proj.bias"), ao, E, R, E, E)
            tgt = norm(ao, tgt, L + ".norm2")
            h = torch.empty(R, FF, dtype=F32, device=dev)
            lin(tgt, E, None, 0, fp(L + ".linear1.weight"), fp(L + ".linear1.bias"), h, FF, R, FF, E, 1)
            lin(h, FF, None, 0, fp(L + ".linear2.weight"), fp(L + ".linear2.bias"), ao, E, R, E, FF)
            tgt = norm(ao, tgt, L + ".norm3")
            norm(tgt, None, "transformer.decoder.norm", y=hs.data_ptr() + 2 * i * R * E, y32=hs32.data_ptr() + 4 * i * R * E)
        return hs32

    def _decoder_coop_launch(self, st, log, kvs, qpos, hs, kpm, B, Q, Lm, lay_n, pdrop, pattn):
        """tuber_decoder_coop_fwd over what the dry run of the decoder's op sequence logged (12 ops per layer: in-proj, attention, out-proj,
        norm1, q-proj, attention, out-proj, norm2, linear1, linear2, norm3, decoder.norm)."""
        import ctypes
        assert len(log) == 12 * lay_n, len(log)
        dev = st.device
        ptr = lambda t: t if isinstance(t, int) else t.data_ptr()
        scratch = []

        def ln(d, with_y=True):
            xh, rs = d["xhat"], d["rstd"]
            if xh is None:                                      # eval: nothing is saved, the kernel still writes somewhere
                xh, rs = torch.empty(B * Q, self.hidden_dim, dtype=BF, device=dev), torch.empty(B * Q, dtype=torch.float32, device=dev)
                scratch.extend((xh, rs))
            return ([d["yptr"]] if with_y else []) + [ptr(xh), ptr(rs)]
        ptrs, salts = [], []
        for i in range(lay_n):
            o = [d for _, d in log[12 * i: 12 * i + 12]]
            kinds = [k for k, _ in log[12 * i: 12 * i + 12]]
            assert kinds == ["in_proj", "attention", "linear", "layer_norm", "in_proj", "attention", "linear", "layer_norm", "linear", "linear",
                             "layer_norm", "layer_norm"], kinds
            w = [o[0]["w"], o[2]["w"], o[4]["w"], o[6]["w"], o[8]["w"], o[9]["w"]]
            b = [o[0]["b"], o[2]["b"], o[4]["b"], o[6]["b"], o[8]["b"], o[9]["b"]]
            lnp = [o[3]["gamma"], o[3]["beta"], o[7]["gamma"], o[7]["beta"], o[10]["gamma"], o[10]["beta"]]
            saved = ([ptr(o[0]["y"]), ptr(o[1]["o"]), ptr(o[1]["lse"]), ptr(o[2]["y"])] + ln(o[3]) +
                     [ptr(o[4]["y"]), ptr(o[5]["o"]), ptr(o[5]["lse"]), ptr(o[6]["y"])] + ln(o[7]) +
                     [ptr(o[8]["y"]), ptr(o[9]["y"])] + ln(o[10]) + ln(o[11], with_y=False))
            ptrs += w + b + lnp + [ptr(kvs[i])] + saved
            salts += [o[1]["salt"], o[3]["salt"], o[5]["salt"], o[7]["salt"], o[8]["salt"], o[10]["salt"]]
        n = lib.query("tuber_decoder_coop_ptrs_per_layer")
        assert len(ptrs) == n * lay_n, (len(ptrs), n)
        P = (ctypes.c_void_p * len(ptrs))(*ptrs)
        S = (ctypes.c_ulonglong * len(salts))(*salts)
        gN = st.flat.data_ptr() + 4 * st.offsets["transformer.decoder.norm.weight"]
        eN = st.flat.data_ptr() + 4 * st.offsets["transformer.decoder.norm.bias"]
        lib.call("tuber_decoder_coop_fwd", P, S, lay_n, qpos, gN, eN, hs, kpm, B, Q, Lm, float(pdrop), float(pattn), st.seed, st.coop_sync)
        del scratch

    # -- forward -----------------------------------------------------------------------------------
    def forward(self, samples):
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_tensor_list(samples)
        st, _ = self.engine()
        dev = st.device
        clips = samples.tensors.to(dev, torch.float32).contiguous()
        mask = samples.mask.to(dev)
        record = self.training and torch.is_grad_enabled()
        logits, logits_b, boxes = _ModelFn.apply(self._anchor, self, clips, mask, record)
        if self.dataset_mode != "ava":
            logits_b = logits_b.unsqueeze(0).repeat(logits.shape[0], 1, 1)
        out = {"pred_logits": logits[-1], "pred_boxes": boxes[-1], "pred_logits_b": logits_b[-1]}
        out["_stacked"] = (logits, logits_b, boxes)      # all decoder layers, contiguous: what the fused criterion consumes
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_logits": a_, "pred_boxes": b_, "pred_logits_b": c_}
                                  for a_, b_, c_ in zip(logits[:-1], boxes[:-1], logits_b[:-1])]
        return out

    def _run(self, clips, mask, record):
        """the whole network on the tape (tape.py); returns (tape, (logits2d, logits_b2d, boxes2d), output shapes)."""
        st, runner = self.engine()
        dev = st.device
        st.refresh(backward=record)
        st.begin_step(self.training)
        tp = T.Tape(st, record)
        E, H = self.hidden_dim, 8
        enc0 = self.transformer.encoder.layers[0]
        on = 1.0 if self.training else 0.0       # dropout is a no-op in eval
        pdrop, pattn = enc0.dropout.p * on, enc0.self_attn.dropout * on

        # ---- backbone (Backbone.forward, backbone_builder.py:59-90) ----
        feat = T.backbone(tp, runner, clips, self.training)                        # [B*T'*hw, 2048] rows (b,t,hw)
        B, Tp, h, w, C = runner.last_shape
        hw = h * w
        strat = self.backbone.temporal_ds_strategy
        Tm = 1                                   # temporal slots of the encoder memory
        if not self.backbone.ds:                 # SINGLE_FRAME: False (backbone_builder.py:70,81-86): no temporal pooling, the encoder
            xs, Tm = feat, Tp                    # sees all T' x h x w tokens (rows (b, t, hw) are already the flatten(2) order)
        elif strat == "avg":
            if Tp != self.backbone.pool_len:
                raise ValueError("TEMPORAL_DS_STRATEGY 'avg' needs T/8 == TEMP_LEN/DS_RATE (got %d vs %d)" % (Tp, self.backbone.pool_len))
            xs = T.gather_sum(tp, feat, (B, 1, hw, Tp, Tp * hw, 0, 1, hw, 1.0 / Tp), (B, Tp, hw, 1, hw, 0, 1, 0, 1.0 / Tp))
        elif strat == "decode":
            xs = self._lstr_pool(tp, feat, B, Tp, hw, on)
        elif strat == "max":
            if Tp != self.backbone.pool_len:
                raise ValueError("TEMPORAL_DS_STRATEGY 'max' needs T/8 == TEMP_LEN/DS_RATE (got %d vs %d)" % (Tp, self.backbone.pool_len))
            xs = T.temporal_max(tp, feat, B, Tp, hw)
        else:                                   # mid-frame slice (backbone_builder.py:79-80)
            xs = T.mid_frame(tp, feat, B, Tp, hw)
        kpm = torch.empty(B, hw, dtype=torch.uint8, device=dev)                       # [B,h*w] key-padding mask (backbone_builder.py:85)
        mk = mask if mask.dtype in (torch.bool, torch.uint8) else mask != 0
        lib.call("tuber_mask_resize", mk.contiguous(), kpm, B, mask.shape[-2], mask.shape[-1], h, w)
        if Tm > 1:                                                                    # mask.unsqueeze(1).repeat(1, T', 1, 1) (:86)
            kpm = kpm[:, None, :].expand(B, Tm, hw).contiguous().view(B, Tm * hw)
        Lm = Tm * hw
        pos = torch.empty(B * Lm, E, dtype=BF, device=dev)
        lib.call("tuber_posenc", kpm, pos, B, Tm, h, w, E)

        # ---- DETR encoder / decoder (transformer.py:49-64) ----
        src = T.linear(tp, xs, "input_proj.weight", "input_proj.bias")             # rows (b, [t,] hw)
        for i in range(self.transformer.encoder.num_layers):
            L = "transformer.encoder.layers.%d" % i
            a = self._mha_self(tp, src, pos, L + ".self_attn", B, Lm, kpm, pattn)
            src = T.layer_norm(tp, a, src, L + ".norm1", drop=pdrop)
            src = T.layer_norm(tp, self._ffn(tp, src, L, pdrop), src, L + ".norm2", drop=pdrop)
        memory = src
        Q = self.query_embed.num_embeddings
        qpos = T.param_rows(tp, "query_embed.weight", B)                            # rows (b, q)
        tgt = torch.zeros(B * Q, E, dtype=BF, device=dev)
        lay_n = self.transformer.decoder.num_layers
        hs = torch.empty(lay_n * B * Q, E, dtype=BF, device=dev)                    # rows (layer, b, q)
        # (the eval precision mode keeps the decoder's residual stream fp32 from LayerNorm to LayerNorm: the launch chain, whose LayerNorm
        #  kernel has that form; the cooperative launch holds its state as bf16 LDS images)
        coop = (not ab.on("no_decoder_coop") and not st.coop_off and (self.training or not ab.eval_fp32_stream()) and lib.query("tuber_decoder_coop_supported", E, H, self.transformer.decoder.layers[0].linear1.out_features, B, Q, lay_n) == 1)
        # eval precision mode: the decoder stack and the box / actor heads in fp32 (csrc/eval_f32.hip) -- a few MFLOP per layer on <= 640 rows
        f32dec = not self.training and ab.eval_fp32_stream() and E // H == 32 and not ab.on("eval_bf16_decoder")
        hs32 = None
        if f32dec:
            hs32 = self._decoder_f32(tp, st, memory, pos, kpm, hs, B, Q, Lm, lay_n, E, H)
            lay_run = 0
        else:
            lay_run = lay_n
        if coop:
            # the decoder stack as ONE cooperative launch (csrc/decoder_coop.hip): the memory-side projections first (they do not depend on
            # the decoder state), then a DRY run of the same op sequence -- it allocates every saved tensor, draws the dropout salts and
            # records the backward closures of the launch chain -- and the fused kernel fills what the dry run allocated
            kvs = []
            for i in range(lay_n):
                P = "transformer.decoder.layers.%d.multihead_attn" % i
                kvs.append(T.in_proj(tp, memory, pos, P + ".in_proj_weight", P + ".in_proj_bias", (E, 3 * E), E))
            tp.dry, tp.dry_log = True, []
        try:
            for i in range(lay_run):
                L = "transformer.decoder.layers.%d" % i
                a = self._mha_self(tp, tgt, qpos, L + ".self_attn", B, Q, None, pattn)
                tgt = T.layer_norm(tp, a, tgt, L + ".norm1", drop=pdrop)
                P = L + ".multihead_attn"
                q = T.in_proj(tp, tgt, qpos, P + ".in_proj_weight", P + ".in_proj_bias", (0, E), E)                  # (tgt + query_pos) W_q
                kv = kvs[i] if coop else T.in_proj(tp, memory, pos, P + ".in_proj_weight", P + ".in_proj_bias", (E, 3 * E), E)          # [(memory + pos) W_k | memory W_v]
                a = T.attention(tp, ((0, 0), (1, 0), (1, E)), (B, H, Q, Lm, (1, Q, 0, 1), (1, Lm, 0, 1)), kpm, pattn, q, kv)
                a = T.linear(tp, a, P + ".out_proj.weight", P + ".out_proj.bias")
                tgt = T.layer_norm(tp, a, tgt, L + ".norm2", drop=pdrop)
                tgt = T.layer_norm(tp, self._ffn(tp, tgt, L, pdrop), tgt, L + ".norm3", drop=pdrop)
                T.layer_norm(tp, tgt, None, "transformer.decoder.norm", out=(hs, i * B * Q, 0))
        finally:
            log, tp.dry, tp.dry_log = tp.dry_log, False, []
        if coop:
            self._decoder_coop_launch(st, log, kvs, qpos, hs, kpm, B, Q, Lm, lay_n, pdrop, pattn)

        # ---- heads (tuber_ava.py:121-125,142) ----
        F32 = torch.float32
        fp = lambda name: st.flat.data_ptr() + 4 * st.offsets[name]      # fp32 master parameter
        if self.dataset_mode == "ava" and f32dec:
            logits_b = torch.empty(lay_n * B * Q, 3, dtype=F32, device=dev)
            lib.call("tuber_linear_f32", hs32, E, None, 0, 0, fp("class_embed_b.weight"), E, fp("class_embed_b.bias"), logits_b, 3, lay_n * B * Q, 3, E, 0)
            lb_shape = (lay_n, B, Q, 3)
        elif self.dataset_mode == "ava":
            logits_b = T.linear(tp, hs, "class_embed_b.weight", "class_embed_b.bias", out_f32=True)
            lb_shape = (lay_n, B, Q, 3)
        else:
            pooled = T.gather_sum(tp, feat, (B, 1, 1, Tp * hw, Tp * hw, 0, 0, 1, 1.0 / (Tp * hw)),
                                  (B, 1, Tp * hw, 1, 1, 0, 0, 0, 1.0 / (Tp * hw)))
            logits_b = T.linear(tp, pooled, "class_embed_b.weight", "class_embed_b.bias", out_f32=True)
            lb_shape = (B, logits_b.shape[1])
        if f32dec:
            Rh = lay_n * B * Q
            x1, x2, boxes = (torch.empty(Rh, E, dtype=F32, device=dev), torch.empty(Rh, E, dtype=F32, device=dev), torch.empty(Rh, 4, dtype=F32, device=dev))
            lib.call("tuber_linear_f32", hs32, E, None, 0, 0, fp("bbox_embed.layers.0.weight"), E, fp("bbox_embed.layers.0.bias"), x1, E, Rh, E, E, 1)
            lib.call("tuber_linear_f32", x1, E, None, 0, 0, fp("bbox_embed.layers.1.weight"), E, fp("bbox_embed.layers.1.bias"), x2, E, Rh, E, E, 1)
            lib.call("tuber_linear_f32", x2, E, None, 0, 0, fp("bbox_embed.layers.2.weight"), E, fp("bbox_embed.layers.2.bias"), boxes, 4, Rh, 4, E, 2)
        else:
            x = T.linear(tp, hs, "bbox_embed.layers.0.weight", "bbox_embed.layers.0.bias", relu=True)
            x = T.linear(tp, x, "bbox_embed.layers.1.weight", "bbox_embed.layers.1.bias", relu=True)
            boxes = T.sigmoid(tp, T.linear(tp, x, "bbox_embed.layers.2.weight", "bbox_embed.layers.2.bias", out_f32=True))

        # ---- class branch (tuber_ava.py:127-141; transformer_layers.py:71-97) ----
        src_c = T.linear(tp, feat, "class_proj.weight", "class_proj.bias")         # rows (b, t, hw)
        R0 = B * Tp * hw
        rep = T.gather_sum(tp, src_c, (lay_n, 1, R0, 1, 0, 0, 1, 0, 1.0), (1, 1, R0, lay_n, 0, 0, 1, R0, 1.0))  # rows (l,b,t,hw)
        LB = lay_n * B
        cl = self.encoder.layers[0]
        pa_c, p1_c, pf_c = cl.self_attn_t.dropout * on, cl.dropout1.p * on, cl.dropout.p * on
        P = "encoder.layers.0"
        cat = torch.empty(lay_n * R0, 2 * E, dtype=BF, device=dev)                  # [t-attention | s-attention] features
        qkv = T.linear(tp, rep, P + ".self_attn_t.in_proj_weight", P + ".self_attn_t.in_proj_bias")
        mp = (1, hw, 0, 1)                       # sequence over hw, batch (lb, t)
        a = T.attention(tp, ((0, 0), (0, E), (0, 2 * E)), (LB * Tp, H, hw, hw, mp, mp), None, pa_c, qkv)
        a = T.linear(tp, a, P + ".self_attn_t.out_proj.weight", P + ".self_attn_t.out_proj.bias")
        T.layer_norm(tp, a, rep, P + ".norm1_t", drop=p1_c, out=(cat, 0, 0))
        qkv = T.linear(tp, rep, P + ".self_attn_s.in_proj_weight", P + ".self_attn_s.in_proj_bias")
        mp = (hw, Tp * hw, 1, hw)                # sequence over t, batch (lb, hw)
        a = T.attention(tp, ((0, 0), (0, E), (0, 2 * E)), (LB * hw, H, Tp, Tp, mp, mp), None, cl.self_attn_s.dropout * on, qkv)
        a = T.linear(tp, a, P + ".self_attn_s.out_proj.weight", P + ".self_attn_s.out_proj.bias")
        T.layer_norm(tp, a, rep, P + ".norm1_s", drop=p1_c, out=(cat, 0, E))
        enc = T.layer_norm(tp, self._ffn(tp, cat, P, pf_c), rep, P + ".norm2", drop=pf_c)
        q = T.linear(tp, hs, "cross_attn.in_proj_weight", "cross_attn.in_proj_bias", rows=(0, E))
        kv = T.linear(tp, enc, "cross_attn.in_proj_weight", "cross_attn.in_proj_bias", rows=(E, 3 * E))
        a = T.attention(tp, ((0, 0), (1, 0), (1, E)), (LB, H, Q, Tp * hw, (1, Q, 0, 1), (1, Tp * hw, 0, 1)), None,
                        self.cross_attn.dropout * on, q, kv)
        q_class = T.linear(tp, a, "cross_attn.out_proj.weight", "cross_attn.out_proj.bias", drop=self.dropout.p * on)
        logits = T.linear(tp, q_class, "class_fc.weight", "class_fc.bias", out_f32=True)
        shapes = ((lay_n, B, Q, logits.shape[1]), lb_shape, (lay_n, B, Q, 4))
        return tp, (logits, logits_b, boxes), shapes

    def _lstr_pool(self, tp, feat, B, Tp, hw, on):
        """TEMPORAL_DS_STRATEGY 'decode' (backbone_builder.py:74-78; transformer_layers.py:380-448): per-pixel one-query
        decoder (d=2048, 8 heads of 256) over the T' temporal slots.  Rows: queries (b, hw); memory (b, t, hw)."""
        E = 2048
        P = "backbone.pool_decoder.layers.0"
        NQ = B * hw
        lay = self.backbone.pool_decoder.layers[0]
        p = lay.self_attn.dropout * on
        pr = lay.dropout1.p * on
        tgt = T.param_rows(tp, "backbone.query_pool.weight", NQ)                    # the same learned query for every pixel
        S = P + ".self_attn"
        q = T.linear(tp, tgt, S + ".in_proj_weight", S + ".in_proj_bias", rows=(0, E))
        kv = T.linear(tp, tgt, S + ".in_proj_weight", S + ".in_proj_bias", rows=(E, 3 * E))
        a = T.attention_wide(tp, q, kv, hw, 1, p)                                   # one key: softmax = 1 (dropout still applies)
        a = T.linear(tp, a, S + ".out_proj.weight", S + ".out_proj.bias")
        tgt = T.layer_norm(tp, a, tgt, P + ".norm1", drop=pr)
        Cx = P + ".multihead_attn"
        q = T.linear(tp, tgt, Cx + ".in_proj_weight", Cx + ".in_proj_bias", rows=(0, E))
        kv = T.linear(tp, feat, Cx + ".in_proj_weight", Cx + ".in_proj_bias", rows=(E, 3 * E))
        a = T.attention_wide(tp, q, kv, hw, Tp, p)
        a = T.linear(tp, a, Cx + ".out_proj.weight", Cx + ".out_proj.bias")
        tgt = T.layer_norm(tp, a, tgt, P + ".norm2", drop=pr)
        tgt = T.layer_norm(tp, self._ffn(tp, tgt, P, pr), tgt, P + ".norm3", drop=pr)
        return T.layer_norm(tp, tgt, None, "backbone.pool_decoder.norm")


class _ModelFn(torch.autograd.Function):
    """The whole network as ONE autograd node: forward runs DETR._run on the tape, backward replays the tape.  The criterion
    (autograd or the fused HIP one) sees ordinary fp32 tensors; ``loss.backward()`` works as in the reference."""

    @staticmethod
    def forward(ctx, anchor, model, clips, mask, record):
        tp, outs, shapes = model._run(clips, mask, record)
        ctx.tp, ctx.outs = tp, outs
        return tuple(o.view(s) for o, s in zip(outs, shapes))

    @staticmethod
    def backward(ctx, *grads):
        tp, outs = ctx.tp, ctx.outs
        seeds = [(o, g.contiguous().view(o.shape) if g is not None else None) for o, g in zip(outs, grads)]
        tp.backward(seeds)
        tp.store.defer.flush()             # every deferred weight-gradient reduction of this pass, one launch
        ctx.tp = ctx.outs = None
        return None, None, None, None, None


def build_model(cfg):
    """models/tuber_ava.py:160-221 -> (model, criterion, postprocessors)."""
    from .criterion import SetCriterion, SetCriterionAVA, PostProcess, PostProcessAVA, build_matcher
    C = cfg.CONFIG
    num_classes = C.DATA.NUM_CLASSES
    backbone = build_backbone(cfg)
    transformer = build_transformer(cfg)
    model = DETR(backbone, transformer, num_classes=num_classes, num_queries=C.MODEL.QUERY_NUM, aux_loss=C.TRAIN.AUX_LOSS,
                 hidden_dim=C.MODEL.D_MODEL, temporal_length=C.MODEL.TEMP_LEN, generate_lfb=C.MODEL.GENERATE_LFB,
                 backbone_name=C.MODEL.BACKBONE_NAME, ds_rate=C.MODEL.DS_RATE, last_stride=C.MODEL.LAST_STRIDE,
                 dataset_mode=C.DATA.DATASET_NAME)
    matcher = build_matcher(cfg)
    weight_dict = {"loss_ce": C.LOSS_COFS.DICE_COF, "loss_bbox": C.LOSS_COFS.BBOX_COF, "loss_giou": C.LOSS_COFS.GIOU_COF,
                   "loss_ce_b": 1}
    if C.TRAIN.AUX_LOSS:
        aux = {}
        for i in range(C.MODEL.DEC_LAYERS - 1):
            aux.update({k + "_%d" % i: v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    losses = ["labels", "boxes"]
    crit_cls = SetCriterionAVA if C.DATA.DATASET_NAME == "ava" else SetCriterion
    criterion = crit_cls(C.LOSS_COFS.WEIGHT, num_classes, num_queries=C.MODEL.QUERY_NUM, matcher=matcher, weight_dict=weight_dict,
                         eos_coef=C.LOSS_COFS.EOS_COF, losses=losses, data_file=C.DATA.DATASET_NAME, evaluation=C.EVAL_ONLY)
    postprocessors = {"bbox": PostProcessAVA() if C.DATA.DATASET_NAME == "ava" else PostProcess()}
    return model, criterion, postprocessors
