"""Drop-in for the reference's ``utils/box_ops.py``."""
from tubelet_transformer_amd.box_ops import *  # noqa: F401,F403
