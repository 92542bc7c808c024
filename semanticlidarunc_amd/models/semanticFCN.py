"""Import-path mirror of the reference's ``models.semanticFCN`` (src/inference_ouster.py:2, src/train_semantics.py).
The implementation lives in ``semanticlidarunc_amd.fpn``."""
from semanticlidarunc_amd.fpn import AttentionModule, SemanticNetworkWithFPN  # noqa: F401

__all__ = ["SemanticNetworkWithFPN", "AttentionModule"]
