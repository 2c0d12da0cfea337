"""Drop-in import path: `from src.models.vggt_qwen3_vlm import VGGTQwen3VLM, VisionLanguageConfig` (what the
reference's train_sft.py:23, qa_inference.py:21, arkit_inference.py:23 and scripts/check_init.py:11 import) now
resolves to the MI355X-native implementation. Nothing else lives here."""
from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig  # noqa: F401

__all__ = ["VGGTQwen3VLM", "VisionLanguageConfig"]
