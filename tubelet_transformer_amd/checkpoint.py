"""Checkpoint / pretrained-weight import and export.

Mirrors ``utils/model_utils.py`` -- ``load_detr_weights`` (:10-36), ``load_model`` (:66-95), ``save_checkpoint`` (:118-134) --
and the Caffe2 ``.mat`` loader of the CSN backbones (``models/backbones/ir_CSN_152.py:213-318``, ``ir_CSN_50.py``).
Everything copies IN PLACE into the existing parameters, so the ParamStore's flat-buffer views stay valid.
The released TubeR checkpoints were saved from a DistributedDataParallel model (keys start with ``module.``); this build's
model is not wrapped, so a leading ``module.`` is accepted on load and (by default) written on save, keeping files
interchangeable with the reference in both directions.
"""
import os

import torch

START_COUNT = {"CSN-152": [0, 3, 11, 47], "CSN-50": [0, 3, 7, 13], "CSN-TEST": [0, 2, 4, 6]}   # ir_CSN_152.py:272 / ir_CSN_50.py:272


def _strip(k):
    return k[7:] if k.startswith("module.") else k


@torch.no_grad()
def _copy_into(model, tensors, verbose=True, what="checkpoint"):
    """copy {name: tensor} into model.state_dict() entries with the same (prefix-stripped) name.  Names the model does not have
    are reported as unused (the reference filters them the same way, utils/model_utils.py:84-85); a name the model HAS with a
    different shape raises like the reference's ``load_state_dict`` does (:89) -- a wrong-NUM_CLASSES / wrong-backbone file
    must not load partially.  Copies are in place (the flat parameter store's views stay valid)."""
    sd = model.state_dict()
    used, unused, bad = [], [], []
    for k, v in tensors.items():
        n = _strip(k)
        if n not in sd:
            unused.append(k)
        elif tuple(sd[n].shape) != tuple(v.shape):
            bad.append("%s: file %s vs model %s" % (n, tuple(v.shape), tuple(sd[n].shape)))
        else:
            used.append((n, v))
    if bad:
        raise RuntimeError("%s: size mismatch for %d tensors:\n  %s" % (what, len(bad), "\n  ".join(bad[:20])))
    for n, v in used:
        sd[n].copy_(torch.as_tensor(v).to(sd[n].device, sd[n].dtype))
    used = [n for n, _ in used]
    missing = [k for k in sd if k not in set(used)]
    if verbose:
        print("%s: loaded %d tensors; unused %d; not found in file %d" % (what, len(used), len(unused), len(missing)))
    store = getattr(model, "_store", None)
    if store is not None and not store.valid():
        raise RuntimeError("%s: parameters were re-allocated while loading (flat store invalid)" % what)
    return used, unused, missing


def load_model(model, cfg, load_fc=True):
    """utils/model_utils.py:66-95: weights-only resume from CONFIG.MODEL.PRETRAINED_PATH (optimizer / epoch are ignored,
    like the reference).  Returns (model, None)."""
    path = cfg.CONFIG.MODEL.PRETRAINED_PATH
    if os.path.isfile(path):
        print("=> loading checkpoint '{}'".format(path))
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        tensors = ckpt["model"] if "model" in ckpt else ckpt.get("state_dict", ckpt)
        if not load_fc:
            tensors = {k: v for k, v in tensors.items() if not _strip(k).startswith("fc.")}
        _copy_into(model, tensors, what="checkpoint")
        print("=> loaded checkpoint '{}' (epoch {})".format(path, ckpt.get("epoch")))
    else:
        print("=> no checkpoint found at '{}'".format(path))
    return model, None


def load_detr_weights(model, pretrain_dir, cfg):
    """utils/model_utils.py:10-36: initialise ``transformer.*``, ``bbox_embed.*`` and the first QUERY_NUM rows of
    ``query_embed.*`` from a DETR checkpoint.  The reference selects entries by their SECOND dotted component (``k.split('.')[1]``)
    and then keeps only keys its DistributedDataParallel-wrapped model has (``k in model_dict``, i.e. ``module.<name>``); this
    build's model is the unwrapped equivalent, so an entry loads iff its key is ``module.<name>`` with ``<name>`` a tensor of the
    model -- a file with any other leading component (``detr.``) matches nothing there and loads nothing here
    (pinned against the reference's own loader: oracle/gen_weight_import_golden.py -> tests/golden/weight_import.json)."""
    ckpt = torch.load(pretrain_dir, map_location="cpu", weights_only=False)
    M = cfg.CONFIG.MODEL
    qsize = M.QUERY_NUM if M.SINGLE_FRAME else M.QUERY_NUM * (M.TEMP_LEN // M.DS_RATE)
    picked, foreign = {}, []
    for k, v in ckpt["model"].items():
        parts = k.split(".")
        if len(parts) < 2 or parts[1] not in ("transformer", "bbox_embed", "query_embed"):
            continue
        if parts[0] != "module":
            foreign.append(k)
            continue
        picked[".".join(parts[1:])] = v[:qsize] if parts[1] == "query_embed" else v
    used, unused, _ = _copy_into(model, picked, what="detr init")
    print("detr unused model layers:", unused + foreign)
    if not used:
        # the reference prints "load pretrain success" here as well (its DDP-wrapped model matches only ``module.`` keys); a silent
        # random-init transformer is the worst outcome of a mis-prefixed file, so say so loudly
        import warnings
        warnings.warn("load_detr_weights: NO tensor of %r was loaded (%d candidate entries, none keyed 'module.<name>' for a tensor of this "
                      "model; first keys: %s) -- the transformer keeps its initialisation" % (pretrain_dir, len(picked) + len(foreign), (foreign + list(picked))[:3]),
                      RuntimeWarning, stacklevel=2)
    print("load pretrain success (%d tensors loaded)" % len(used))
    return model


def save_checkpoint(cfg, epoch, model, max_accuracy, optimizer, lr_scheduler, ddp_prefix=True):
    """utils/model_utils.py:118-134: {model, optimizer, lr_scheduler, max_accuracy, epoch, config} ->
    {BASE_PATH}/{EXP_NAME}/{SAVE_DIR}/ckpt_epoch_{e}.pth."""
    sd = model.state_dict()
    if ddp_prefix:
        sd = {"module." + k: v for k, v in sd.items()}
    save_state = {"model": {k: v.detach().cpu() for k, v in sd.items()},
                  "optimizer": optimizer.state_dict() if optimizer is not None else None,
                  "lr_scheduler": lr_scheduler.state_dict() if lr_scheduler is not None else None,
                  "max_accuracy": max_accuracy, "epoch": epoch,
                  "config": cfg.to_dict() if hasattr(cfg, "to_dict") else cfg}
    d = os.path.join(cfg.CONFIG.LOG.BASE_PATH, cfg.CONFIG.LOG.EXP_NAME, cfg.CONFIG.LOG.SAVE_DIR)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "ckpt_epoch_%d.pth" % epoch)
    print("Saving model at epoch %d to %s" % (epoch, d))
    torch.save(save_state, path)
    return path


@torch.no_grad()
def load_csn_mat(body, pretrain_path, backbone_name, tune_point=4, verbose=True):
    """Caffe2 ir-CSN weights (``scipy.io.loadmat``) into the ResNeXt body -- ir_CSN_152.py:242-318:
    ``conv1_w``, ``conv1_spatbn_relu_{s,b,rm,riv}``, ``comp_{i}_conv_{1,3,4}_w``, ``comp_{i}_spatbn_{1,3,4}_{s,b,rm,riv}``,
    ``shortcut_projection_{i}_w``, ``shortcut_projection_{i}_spatbn_{s,b,rm,riv}`` with block index i continuing over the
    stages (start offsets [0,3,11,47] / [0,3,7,13]).  Freezing follows the reference: the stem when tune_point > 1 and
    stage s when tune_point > s + 2 (``build_CSN`` passes tune_point = 4: stem, layer1, layer2 frozen)."""
    import scipy.io as sio
    w = sio.loadmat(pretrain_path)
    left = {k for k in w if not k.startswith("__")}

    def put(dst, name, shape=None):
        v = torch.from_numpy(w[name]).float()
        v = v.reshape(shape if shape is not None else dst.shape)
        assert tuple(v.shape) == tuple(dst.shape), (name, tuple(v.shape), tuple(dst.shape))
        dst.copy_(v)
        left.discard(name)

    def put_bn(bn, name):
        put(bn.weight, name + "_s", (-1,))
        put(bn.bias, name + "_b", (-1,))
        put(bn.running_mean, name + "_rm", (-1,))
        put(bn.running_var, name + "_riv", (-1,))

    put(body.conv1.weight, "conv1_w")
    put_bn(body.bn1, "conv1_spatbn_relu")
    if tune_point > 1:
        body.conv1.weight.requires_grad = False
        for p in body.bn1.parameters():
            p.requires_grad = False
    stages = [body.layer1, body.layer2, body.layer3, body.layer4]
    for s, stage in enumerate(stages):
        count = START_COUNT[backbone_name][s]
        for blk in stage:
            for j, (conv, bn) in zip((1, 3, 4), ((blk.conv1, blk.bn1), (blk.conv3, blk.bn3), (blk.conv4, blk.bn4))):
                put(conv.weight, "comp_%d_conv_%d_w" % (count, j))
                put_bn(bn, "comp_%d_spatbn_%d" % (count, j))
            if blk.down_sample is not None:
                put(blk.down_sample[0].weight, "shortcut_projection_%d_w" % count)
                put_bn(blk.down_sample[1], "shortcut_projection_%d_spatbn" % count)
            count += 1
        if tune_point > s + 2:
            for p in stage.parameters():
                p.requires_grad = False
    if verbose:
        rest = [k for k in left if not any(t in k for t in ("momentum", "model_iter", "lr"))]
        print("load pretrain model " + pretrain_path + "; unconsumed entries:", sorted(rest)[:8])
    return body
