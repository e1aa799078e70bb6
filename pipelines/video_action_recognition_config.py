"""Drop-in for ``pipelines/video_action_recognition_config.py:get_cfg_defaults`` (yacs-free)."""
from tubelet_transformer_amd.config import CfgNode, get_cfg_defaults  # noqa: F401
