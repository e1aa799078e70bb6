"""Import the UNMODIFIED reference in the build container (never on the GPU box).

Harness glue only: three in-process ``sys.modules`` shims
(torchvision, cv2, yacs-free cfg) so that ``models.tuber_ava.build_model`` from
``/root/reference`` can be called.  Used by ``oracle/gen_golden.py`` to pin the
oracle and to generate the committed vectors under ``tests/golden/``.
Nothing here is imported by the product or by ``-m gpu`` tests.
"""
import contextlib
import io
import os
import sys
import types

import torch
import yaml

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "models"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims():
    if "torchvision" not in sys.modules or not hasattr(sys.modules["torchvision"], "_tuber_shim"):
        tv = _mod("torchvision", __version__="0.15.0", _tuber_shim=True)
        _mod("torchvision.ops")
        _mod("torchvision.models")
        _mod("torchvision.models.video")
        _mod("torchvision.ops.boxes", box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
        _mod("torchvision.ops.misc", interpolate=torch.nn.functional.interpolate)
        _mod("torchvision.models._utils", IntermediateLayerGetter=object)
        _mod("torchvision.models.video.resnet", VideoResNet=object)
        tv.ops = sys.modules["torchvision.ops"]
        tv.ops.misc = sys.modules["torchvision.ops.misc"]
        tv.ops.boxes = sys.modules["torchvision.ops.boxes"]
        _mod("cv2")


class _Node(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _to_node(d):
    return _Node({k: _to_node(v) for k, v in d.items()}) if isinstance(d, dict) else d


def ref_cfg(yaml_name, **model_overrides):
    """Attribute-dict cfg from the REFERENCE's own YAML (stands in for the yacs node)."""
    cfg = _to_node(yaml.safe_load(open(os.path.join(REF, "configuration", yaml_name))))
    for k in ("LR", "MIN_LR", "LR_BACKBONE", "W_DECAY", "WARMUP_START_LR"):
        if k in cfg.CONFIG.TRAIN:
            cfg.CONFIG.TRAIN[k] = float(cfg.CONFIG.TRAIN[k])
    return cfg


@contextlib.contextmanager
def reference_on_path():
    """Temporarily put /root/reference first on sys.path and hide this repo's own
    ``models`` / ``utils`` / ``pipelines`` drop-in packages so the reference's resolve."""
    install_shims()
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k.split(".")[0] in ("models", "utils", "pipelines")}
    sys.path.insert(0, REF)
    try:
        yield
    finally:
        sys.path.remove(REF)
        for k in list(sys.modules):
            if k.split(".")[0] in ("models", "utils", "pipelines"):
                del sys.modules[k]
        sys.modules.update(saved)


def build_reference(cfg):
    """(model, criterion, postprocessors) from the reference's build_model, banner prints muted."""
    with reference_on_path():
        from models.tuber_ava import build_model
        with contextlib.redirect_stdout(io.StringIO()):
            out = build_model(cfg)
    return out


def zero_dropout(model):
    """Make train-mode deterministic: every dropout probability -> 0."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0
    return model
