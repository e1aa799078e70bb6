"""The reference's ``models/tuber_jhmdb.py`` is un-importable (it imports a module that does not exist,
models/tuber_jhmdb.py:20) and both JHMDB scripts use ``models.tuber_ava.build_model``.
This alias exposes the intended surface (:407-448): the same DETR with ``dataset_mode='jhmdb'``."""
from tubelet_transformer_amd.tuber import DETR, build_model  # noqa: F401
