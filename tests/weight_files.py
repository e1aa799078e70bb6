"""Seeded synthetic weight FILES in the formats the reference's loaders read -- test infrastructure shared
by ``oracle/gen_weight_import_golden.py`` (which feeds them to the REFERENCE's own loaders in the build container and records what
those produce) and by the tests (which feed the same files to ``tubelet_transformer_amd.checkpoint`` and compare):

  * a Caffe2 ir-CSN ``.mat`` (names of models/backbones/ir_CSN_152.py:242-318 / ir_CSN_50.py, block counter running over the stages);
  * a TubeR checkpoint saved from a DistributedDataParallel model (``module.`` prefix), with one entry the model does not have and
    one model entry missing (utils/model_utils.py:66-95 filters both);
  * DETR initialisation files whose keys carry one leading component (utils/model_utils.py:10-36): ``module.`` (what a
    DDP-wrapped TubeR matches) and ``detr.`` (which it does not match: nothing is loaded).
Only shapes come from the model handed in; values come from ``numpy.random.default_rng(seed)`` in a fixed order."""
import zlib

import numpy as np
import torch

BLOCKS = {"CSN-152": [3, 8, 36, 3], "CSN-50": [3, 4, 6, 3]}


def crc(t):
    return zlib.crc32(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()) & 0xFFFFFFFF


def csn_mat_dict(backbone_name, seed):
    """{caffe2 name: float32 array} for a full ir-CSN body; shapes from the architecture (ir_CSN_152.py:36-68,109-135)."""
    rng = np.random.default_rng(seed)
    mat = {}

    def conv(name, shape):
        fan = int(np.prod(shape[1:]))
        mat[name] = (rng.standard_normal(shape) / np.sqrt(fan)).astype(np.float32)

    def bn(name, c):
        mat[name + "_s"] = (1.0 + 0.1 * rng.standard_normal((c,))).astype(np.float32)
        mat[name + "_b"] = (0.1 * rng.standard_normal((c,))).astype(np.float32)
        mat[name + "_rm"] = (0.1 * rng.standard_normal((c,))).astype(np.float32)
        mat[name + "_riv"] = (1.0 + 0.2 * rng.random((c,))).astype(np.float32)
    conv("conv1_w", (64, 3, 3, 7, 7))
    bn("conv1_spatbn_relu", 64)
    count, cin = 0, 64
    for planes, nblk in zip((64, 128, 256, 512), BLOCKS[backbone_name]):
        for b in range(nblk):
            conv("comp_%d_conv_1_w" % count, (planes, cin, 1, 1, 1))
            bn("comp_%d_spatbn_1" % count, planes)
            conv("comp_%d_conv_3_w" % count, (planes, 1, 3, 3, 3))
            bn("comp_%d_spatbn_3" % count, planes)
            conv("comp_%d_conv_4_w" % count, (planes * 4, planes, 1, 1, 1))
            bn("comp_%d_spatbn_4" % count, planes * 4)
            if b == 0:
                conv("shortcut_projection_%d_w" % count, (planes * 4, cin, 1, 1, 1))
                bn("shortcut_projection_%d_spatbn" % count, planes * 4)
            cin = planes * 4
            count += 1
    # entries the loaders must ignore (the released files carry solver state next to the weights)
    mat["last_out_L400_w"] = rng.standard_normal((400, 2048)).astype(np.float32)
    mat["last_out_L400_b"] = rng.standard_normal((400,)).astype(np.float32)
    mat["model_iter"] = np.array([12345.0], np.float32)
    mat["conv1_w_momentum"] = np.zeros((1,), np.float32)
    return mat


def write_csn_mat(path, backbone_name, seed):
    import scipy.io as sio
    sio.savemat(path, csn_mat_dict(backbone_name, seed))
    return path


def _seeded_like(state_dict, seed):
    out = {}
    for i, (k, v) in enumerate(state_dict.items()):
        g = torch.Generator().manual_seed(seed * 100003 + i)
        if v.dtype.is_floating_point:
            out[k] = torch.randn(v.shape, generator=g).to(v.dtype) * 0.05
        else:
            out[k] = torch.full_like(v, 7)
    return out


def write_tuber_checkpoint(path, model_state_dict, seed, drop="class_fc.bias"):
    """DDP-saved TubeR checkpoint: every key ``module.<name>``; ``drop`` is left out (a 'not found' layer), one foreign key added."""
    sd = {"module." + k: v for k, v in _seeded_like(model_state_dict, seed).items() if k != drop}
    sd["module.some_head_of_another_experiment.weight"] = torch.ones(3, 3)
    torch.save({"model": sd, "epoch": 7, "max_accuracy": 0.0}, path)
    return path


def write_detr_checkpoint(path, model_state_dict, seed, prefix, rows=100):
    """DETR initialisation file: ``<prefix>.transformer.*``, ``<prefix>.bbox_embed.*``, ``<prefix>.query_embed.weight`` with ``rows``
    rows (DETR's 100 object queries), plus DETR-only tensors the TubeR model has no counterpart for."""
    vals = _seeded_like(model_state_dict, seed)
    sd = {}
    for k, v in vals.items():
        if k.startswith(("transformer.", "bbox_embed.")):
            sd[prefix + "." + k] = v
    g = torch.Generator().manual_seed(seed + 17)
    sd[prefix + ".query_embed.weight"] = torch.randn(rows, model_state_dict["query_embed.weight"].shape[1], generator=g)
    sd[prefix + ".class_embed.weight"] = torch.randn(92, 256, generator=g)
    sd[prefix + ".backbone.0.body.conv1.weight"] = torch.randn(64, 3, 7, 7, generator=g)
    torch.save({"model": sd}, path)
    return path


def snapshot(model):
    """{name: [crc32 of the tensor bytes, requires_grad or None for buffers]} of a model's full state"""
    req = {n: bool(p.requires_grad) for n, p in model.named_parameters()}
    return {k: [crc(v), req.get(k)] for k, v in model.state_dict().items()}
