"""Minimal yacs-compatible config node for the TubeR hot path.

The reference drives everything from a yacs ``CfgNode`` built by
``pipelines/video_action_recognition_config.py:5-222`` (``get_cfg_defaults``)
and ``cfg.merge_from_file(yaml)`` (``train_tuber_ava.py:100-101``).  yacs is not
installed in this image, so this module provides the subset of its behaviour the
hot path relies on: attribute access, ``merge_from_file``, ``merge_from_other``,
``clone``, ``dump``, ``freeze``/``defrost``, ``new_allowed`` sub-trees, and
yacs' habit of ``literal_eval``-ing YAML strings (so ``LR: 1e-4`` becomes a float).
"""
import ast
import copy

import yaml


class CfgNode(dict):
    _FROZEN = "__frozen__"
    _NEW_ALLOWED = "__new_allowed__"

    def __init__(self, init=None, new_allowed=False):
        super().__init__()
        self.__dict__[CfgNode._FROZEN] = False
        self.__dict__[CfgNode._NEW_ALLOWED] = new_allowed
        for k, v in (init or {}).items():
            self[k] = CfgNode(v, new_allowed=new_allowed) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute access -------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode._FROZEN]:
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    # yacs API -----------------------------------------------------------
    def is_frozen(self):
        return self.__dict__[CfgNode._FROZEN]

    def is_new_allowed(self):
        return self.__dict__[CfgNode._NEW_ALLOWED]

    def _set_frozen(self, flag):
        self.__dict__[CfgNode._FROZEN] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode(new_allowed=self.is_new_allowed())
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v))
                for k, v in self.items()}

    def dump(self, **kwargs):
        return yaml.safe_dump(self.to_dict(), **kwargs)

    def merge_from_file(self, cfg_filename):
        with open(cfg_filename, "r") as f:
            loaded = yaml.safe_load(f)
        self.merge_from_other_cfg(loaded)

    def merge_from_other_cfg(self, other):
        _merge(other, self, [])

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0, "Override list has odd length: {}".format(cfg_list)
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = _decode(v)


def _decode(v):
    """yacs ``_decode_cfg_value``: dict -> node, str -> literal_eval if it parses."""
    if isinstance(v, dict):
        return CfgNode(v, new_allowed=True)
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _merge(src, dst, path):
    for k, v in src.items():
        full = ".".join(path + [k])
        v = _decode(v) if not isinstance(v, dict) else v
        if k in dst:
            if isinstance(dst[k], CfgNode) and isinstance(v, dict):
                _merge(v, dst[k], path + [k])
            else:
                if isinstance(dst[k], tuple) and isinstance(v, list):
                    v = tuple(v)
                if isinstance(dst[k], float) and isinstance(v, int):
                    v = float(v)
                dst[k] = CfgNode(v, new_allowed=True) if isinstance(v, dict) else v
        elif dst.is_new_allowed():
            dst[k] = CfgNode(v, new_allowed=True) if isinstance(v, dict) else v
        else:
            raise KeyError("Non-existent config key: {}".format(full))


def get_cfg_defaults():
    """Defaults for the keys the TubeR hot path reads.

    Mirrors the relevant subset of ``pipelines/video_action_recognition_config.py``
    (DDP_CONFIG ``:11-31``, CONFIG.* nodes with ``new_allowed=True`` ``:37-39,105,178,202``)
    plus the hot-path keys every published YAML supplies, with the values of
    ``configuration/TubeR_CSN152_AVA21.yaml`` as defaults so a partial YAML still builds.
    """
    C = CfgNode()
    C.DDP_CONFIG = CfgNode(dict(
        WORLD_SIZE=1, WORLD_RANK=0, GPU_WORLD_SIZE=8, GPU_WORLD_RANK=0,
        DIST_URL="tcp://127.0.0.1:10001", WOLRD_URLS=["127.0.0.1"], AUTO_RANK_MATCH=True,
        DIST_BACKEND="nccl", GPU=0, DISTRIBUTED=True))
    cfg = CfgNode(new_allowed=True)
    cfg.EVAL_ONLY = False
    cfg.TWO_STREAM = False
    cfg.USE_LFB = False
    cfg.USE_LOCATION = False
    cfg.TRAIN = CfgNode(dict(
        START_EPOCH=0, EPOCH_NUM=20, BATCH_SIZE=2, LR=1e-4, MIN_LR=1e-5, LR_BACKBONE=1e-5,
        W_DECAY=1e-4, LR_POLICY="step", AUX_LOSS=True), new_allowed=True)
    cfg.VAL = CfgNode(dict(FREQ=2, BATCH_SIZE=1), new_allowed=True)
    cfg.DATA = CfgNode(dict(
        DATASET_NAME="ava", NUM_CLASSES=80, IMG_SIZE=256, TEMP_LEN=32, FRAME_RATE=2), new_allowed=True)
    cfg.MODEL = CfgNode(dict(
        NAME="", SINGLE_FRAME=True, BACKBONE_NAME="CSN-152", TEMPORAL_DS_STRATEGY="avg", LAST_STRIDE=False,
        GENERATE_LFB=False, ENC_LAYERS=6, DEC_LAYERS=6, D_MODEL=256, NHEAD=8, DIM_FEEDFORWARD=2048,
        QUERY_NUM=15, NORMALIZE_BEFORE=False, DROPOUT=0.1, DS_RATE=8, TEMP_LEN=32, PRETRAINED=False,
        PRETRAIN_BACKBONE_DIR="", PRETRAIN_TRANSFORMER_DIR="", PRETRAINED_PATH="", LOAD=False, LOAD_FC=True),
        new_allowed=True)
    cfg.MATCHER = CfgNode(dict(COST_CLASS=12, COST_BBOX=5, COST_GIOU=2, BNY_LOSS=True, BEFORE=False),
                          new_allowed=True)
    cfg.LOSS_COFS = CfgNode(dict(
        MASK_COF=1, DICE_COF=12, BBOX_COF=5, GIOU_COF=2, EOS_COF=0.1, WEIGHT=10, WEIGHT_CHANGE=1000,
        LOSS_CHANGE_COF=2, CLIPS_MAX_NORM=0.1), new_allowed=True)
    cfg.LOG = CfgNode(dict(BASE_PATH="", EXP_NAME="use_time", LOG_DIR="tb_log", SAVE_DIR="checkpoints",
                           EVAL_DIR="", SAVE_FREQ=1, RES_DIR="tmp"), new_allowed=True)
    C.CONFIG = cfg
    return C


def load_cfg(path):
    cfg = get_cfg_defaults()
    cfg.merge_from_file(path)
    return cfg
