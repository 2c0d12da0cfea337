"""Drop-in import path for `from src.models.projector_perceiver import PerceiverConfig[, PerceiverProjector]`
(reference: train_sft.py:22, qa_inference.py:20, eval_3dqa.py:11): the MI355X-native projector."""
from vggt_qwen3_amd.perceiver import PerceiverConfig, PerceiverLayer, PerceiverProjector  # noqa: F401

__all__ = ["PerceiverConfig", "PerceiverLayer", "PerceiverProjector"]
