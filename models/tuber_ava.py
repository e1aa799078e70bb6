"""Drop-in for the reference's ``models/tuber_ava.py``: ``from models.tuber_ava import build_model`` keeps working
(train_tuber_ava.py:9, eval_tuber_ava.py:9, train_tuber_jhmdb.py:9, eval_tuber_jhmdb.py:9).  The implementation is the
MI355X HIP path in ``tubelet_transformer_amd``."""
from tubelet_transformer_amd.tuber import DETR, build_model  # noqa: F401
from tubelet_transformer_amd.criterion import (SetCriterion, SetCriterionAVA, PostProcess, PostProcessAVA)  # noqa: F401
from tubelet_transformer_amd.tuber import MLP  # noqa: F401
