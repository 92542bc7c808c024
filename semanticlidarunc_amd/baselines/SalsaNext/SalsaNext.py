"""Import-path mirror of the reference's ``baselines.SalsaNext.SalsaNext`` (train_semantics.py:154).
The implementation lives in ``semanticlidarunc_amd.salsanext``."""
from semanticlidarunc_amd.salsanext import ResBlock, ResContextBlock, SalsaNext, UpBlock  # noqa: F401

__all__ = ["SalsaNext", "ResContextBlock", "ResBlock", "UpBlock"]
