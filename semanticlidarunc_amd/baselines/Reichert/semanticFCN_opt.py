"""Import-path mirror of the reference's ``baselines.Reichert.semanticFCN_opt`` (train_semantics.py:134).
The implementation lives in ``semanticlidarunc_amd.fpn_opt``."""
from semanticlidarunc_amd.fpn_opt import GN, SemanticNetworkWithFPN, SpatialAttention, UpsampleBlock  # noqa: F401

__all__ = ["SemanticNetworkWithFPN", "UpsampleBlock", "SpatialAttention", "GN"]


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
