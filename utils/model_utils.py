"""Drop-in for the reference's ``utils/model_utils.py`` (deploy_model, load_model, load_detr_weights, save_checkpoint)."""
from tubelet_transformer_amd.checkpoint import load_detr_weights, load_model, save_checkpoint  # noqa: F401
from tubelet_transformer_amd.training import deploy_model  # noqa: F401
