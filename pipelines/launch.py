"""Drop-in for the reference's ``pipelines/launch.py`` (``spawn_workers``, ``main_worker``)."""
from tubelet_transformer_amd.launch import get_local_ip_and_match, main_worker, spawn_workers  # noqa: F401
