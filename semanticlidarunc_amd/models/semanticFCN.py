"""Import-path mirror of the reference's ``models.semanticFCN`` (src/inference_ouster.py:2, src/train_semantics.py).
The implementation lives in ``semanticlidarunc_amd.fpn``."""
from semanticlidarunc_amd.fpn import AttentionModule, SemanticNetworkWithFPN  # noqa: F401

__all__ = ["SemanticNetworkWithFPN", "AttentionModule"]


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
