"""Drop-in for the hot-path subset of the reference's ``utils/misc.py``."""
from tubelet_transformer_amd.misc import *  # noqa: F401,F403
from tubelet_transformer_amd.misc import NestedTensor, nested_tensor_from_tensor_list, collate_fn  # noqa: F401
