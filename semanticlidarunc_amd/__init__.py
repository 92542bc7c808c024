"""MI355X-native range-image segmentation + uncertainty path (drop-in for the hot path of
kav-institute/SemanticLiDARUnc).  See DESIGN.md / INTEGRATION.md at the repo root.

Sub-packages ``baselines``, ``utils``, ``losses``, ``models`` and ``metrics`` mirror the reference's
``src/`` import paths (put this directory on ``sys.path`` ahead of the reference's ``src/`` and
``train_semantics.py`` / ``inference_ouster.py`` pick the HIP implementations up unchanged).
"""
__version__ = "0.1.0"

from ._lib import SluError, load as load_library  # noqa: F401
