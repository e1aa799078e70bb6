"""Drop-in alias package.  The sub-modules this repository provides (the hot path and its callers) are found here first; any other
sub-module of the reference's package of the same name (``utils.utils``, ``utils.lr_scheduler``, ``pipelines.image_classification_config``,
``models.detr.util`` ...) resolves from the next ``sys.path`` entry that holds such a package, so the reference's entry scripts import
unchanged with this repository first on ``PYTHONPATH`` (train_tuber_ava.py:9-16, eval_tuber_ava.py:9-15, train_tuber_jhmdb.py:9-16)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
