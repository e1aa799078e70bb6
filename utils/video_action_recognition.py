"""Drop-in for the reference's ``utils/video_action_recognition.py`` entry points used by train_tuber_*.py / eval_tuber_*.py."""
from tubelet_transformer_amd.evaluation import validate_tuber_detection, validate_tuber_ucf_detection  # noqa: F401
from tubelet_transformer_amd.training import train_tuber_detection  # noqa: F401
