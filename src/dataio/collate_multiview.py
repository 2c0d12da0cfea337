"""Import-path shim: `from src.dataio.collate_multiview import MultiViewCollator` resolves to the MI355X batch builder."""
from vggt_qwen3_amd.collate import MultiViewCollator, build_default_transform  # noqa: F401
