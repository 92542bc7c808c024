"""Import-path mirror of the reference's ``baselines.SalsaNext.SalsaNext`` (train_semantics.py:154).
The implementation lives in ``semanticlidarunc_amd.salsanext``."""
from semanticlidarunc_amd.salsanext import ResBlock, ResContextBlock, SalsaNext, UpBlock  # noqa: F401

__all__ = ["SalsaNext", "ResContextBlock", "ResBlock", "UpBlock"]


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
