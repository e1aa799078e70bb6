"""Deterministic synthetic weights, clips and targets.

There are no checkpoints or datasets in this image, so parity and throughput
runs use:

* **name-hashed weights** -- every tensor of a ``state_dict`` is filled from a CPU
  generator seeded with ``crc32(name)``, so the reference model (imported only when
  generating golden vectors), the CPU oracle and the HIP model get bit-identical
  weights without sharing construction order;
* **synthetic clips** -- ``randn`` (ImageNet-normalised frames are ~N(0,1));
* **synthetic targets** following the target-dict contract of
  ``datasets/ava_frame.py:112-128`` / ``datasets/jhmdb_frame.py:170-189``.
"""
import zlib

import torch


def _gen(name, salt=0):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


# "spread" fixture (round 5): at plain name-hashed weights the DETR decoder's queries are near-copies of each other -- the random
# encoder averages its tokens into one common vector (memory diversity over tokens 0.08), every query attends almost uniformly, and
# the 15 tubelet queries' boxes differ by <= 1.7e-3, their actor probabilities by 0.06: below bf16 noise, so no bf16 execution can
# reproduce the reference's Hungarian assignment or move a post-processing gate.  These gains (found by measurement on the fp32
# oracle, oracle/gen_golden.py) keep the tokens and the queries apart: identity-dominated encoder layers, sharper attention, larger
# query embeddings, wider actor / box heads.  With them the queries' decoder states differ by ~30 % of their norm (was 0.3 %).
SPREAD_GAINS = {"encoder_residual": 0.1,      # transformer.encoder.layers.*.{self_attn.out_proj, linear2}.{weight, bias}
                "attention_qk": 2.0,          # q / k rows of every transformer.* in_proj_{weight, bias}: scores x 4
                "query_embed": 3.0,
                "class_embed_b": 3.0,         # actor logits: p_b spreads over ~[0.05, 0.95]
                "bbox_last": 2.0}             # bbox_embed.layers.2.weight: boxes spread over >= 0.05 without saturating the sigmoid


def _spread_gain(key, v):
    g = SPREAD_GAINS
    if key.startswith("transformer.encoder.") and key.rsplit(".", 1)[0].endswith(("self_attn.out_proj", "linear2")):
        return v * g["encoder_residual"]
    if key.startswith("transformer.") and key.endswith(("in_proj_weight", "in_proj_bias")):
        v = v.clone()
        v[: 2 * v.shape[0] // 3] *= g["attention_qk"]
        return v
    if key == "query_embed.weight":
        return v * g["query_embed"]
    if key == "class_embed_b.weight":
        return v * g["class_embed_b"]
    if key == "bbox_embed.layers.2.weight":
        return v * g["bbox_last"]
    return v


@torch.no_grad()
def name_hashed_state(state_dict, salt=0, residual_gain=None, spread=False):
    """Return ``{name: tensor}`` with deterministic values for every entry of ``state_dict``.

    Rules (by leaf name): BN/LN ``weight`` ~ 1 + 0.1 N, ``bias`` ~ 0.1 N (0.02 N for
    conv/linear biases), ``running_mean`` ~ 0.1 N, ``running_var`` ~ 1 + 0.2 U,
    ``num_batches_tracked`` = 0, embeddings ~ N, every >=2-D weight ~ N(0, 1/fan_in).
    A leading ``module.`` (DDP prefix) is ignored so wrapped and bare models agree.

    ``residual_gain``: scale every ``bn4.weight`` -- the last BatchNorm of each residual branch (ir_CSN_152.py:64,82-84) -- by this
    factor.  At random weights a 50-bottleneck training-mode-BatchNorm body amplifies bf16 rounding through ReLU-mask flips until even
    an ideally-accumulated bf16 execution decorrelates from fp32; with identity-dominated blocks (gain ~0.05-0.1) the deep gradient
    stays well-conditioned, which is what a parity test at real depth needs (tests/test_fullsize_gpu.py).

    ``spread``: apply ``SPREAD_GAINS`` (non-degenerate tubelet queries; see there).
    """
    out = {}
    for name, t in state_dict.items():
        key = name[7:] if name.startswith("module.") else name
        g = _gen(key, salt)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            v = torch.zeros_like(t)
        elif leaf == "running_mean":
            v = 0.1 * torch.randn(t.shape, generator=g)
        elif leaf == "running_var":
            v = 1.0 + 0.2 * torch.rand(t.shape, generator=g)
        elif leaf == "empty_weight":
            v = t.clone()
        elif t.dim() >= 2:
            if "query_embed" in key or "query_pool" in key:
                v = torch.randn(t.shape, generator=g)
            else:
                fan_in = t[0].numel()
                v = torch.randn(t.shape, generator=g) * (1.0 / fan_in) ** 0.5
        elif leaf == "weight":            # 1-D weight: BatchNorm / LayerNorm gamma
            v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
            if residual_gain is not None and key.endswith(".bn4.weight"):
                v = v * float(residual_gain)
        elif leaf in ("bias", "in_proj_bias"):
            is_norm = any(s in key for s in (".bn", "norm", "down_sample.1"))
            v = (0.1 if is_norm else 0.02) * torch.randn(t.shape, generator=g)
        else:
            v = 0.02 * torch.randn(t.shape, generator=g)
        if spread:
            v = _spread_gain(key, v)
        out[name] = v.to(t.dtype)
    return out


@torch.no_grad()
def load_name_hashed(module, salt=0, residual_gain=None, spread=False):
    """Fill ``module``'s parameters and buffers in place with name-hashed values."""
    sd = module.state_dict()
    vals = name_hashed_state(sd, salt, residual_gain, spread)
    for k, t in sd.items():
        t.copy_(vals[k].to(t.device))
    return module


def synthetic_clips(batch, t, h, w, seed=1234, device="cpu", sizes=None):
    """``batch`` clips (3,t,h,w) ~ N(0,1).  ``sizes`` = per-clip (h,w) for ragged batches."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    if sizes is None:
        return torch.randn(batch, 3, t, h, w, generator=g).to(device)
    return [torch.randn(3, t, hh, ww, generator=g).to(device) for hh, ww in sizes]


def structured_clips(batch, t, h, w, seed=1234, amp=2.0, device="cpu"):
    """Clips with spatial / temporal structure: 0.5 N(0,1) pixel noise + ``amp`` x a trilinearly upsampled coarse random field (one
    value per 16 x 16 pixel cell and 8 frames).  i.i.d. noise gives every backbone position statistically the same content, so
    the feature map (and with it what the queries can attend to) barely varies over positions; this does."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    base = torch.randn(batch, 3, t, h, w, generator=g)
    coarse = torch.randn(batch, 3, 4, max(2, h // 16), max(2, w // 16), generator=g)
    low = torch.nn.functional.interpolate(coarse, size=(t, h, w), mode="trilinear", align_corners=False)
    return (0.5 * base + amp * low).to(device)


def synthetic_targets(batch, dataset="ava", num_classes=80, seed=4321, device="cpu", hw=(256, 340),
                      boxes_per_clip=None):
    """Target dicts with the keys the criterion reads.

    AVA: ``boxes`` float32 [N,5] = (key_t=16, cx, cy, w, h) in [0,1]; ``labels`` float32 [N,80]
    multi-hot.  JHMDB: ``labels`` int64 [N], ``vis`` int64 [1], ``key_pos`` int64 scalar (=16).
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out = []
    for b in range(batch):
        if dataset == "ava":
            n = int(torch.randint(1, 4, (1,), generator=g)) if boxes_per_clip is None else boxes_per_clip[b]
        else:
            n = 1
        cxcy = 0.3 + 0.4 * torch.rand(n, 2, generator=g)
        wh = 0.1 + 0.2 * torch.rand(n, 2, generator=g)
        boxes = torch.cat([torch.full((n, 1), 16.0), cxcy, wh], dim=1)
        t = {"boxes": boxes.to(device),
             "size": torch.tensor(list(hw), dtype=torch.int64, device=device),
             "orig_size": torch.tensor(list(hw), dtype=torch.int64, device=device),
             "area": (wh[:, 0] * wh[:, 1] * hw[0] * hw[1]).to(device)}
        if dataset == "ava":
            lab = (torch.rand(n, num_classes, generator=g) < 0.05).float()
            lab[:, 11] = 1.0
            t["labels"] = lab.to(device)
        else:
            t["labels"] = torch.randint(0, num_classes, (n,), generator=g).to(device)
            t["vis"] = torch.ones(1, dtype=torch.int64, device=device)
            t["key_pos"] = torch.tensor(16, dtype=torch.int64, device=device)
        out.append(t)
    return out


def zero_dropout(model):
    """Deterministic train mode for parity runs: nn.Dropout.p and nn.MultiheadAttention.dropout -> 0."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(getattr(m, "dropout", None), float):
            m.dropout = 0.0
    return model
