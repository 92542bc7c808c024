"""Drop-in plumbing for the sub-packages that mirror the reference's import paths (``utils``, ``models``, ``metrics``,
``losses``, ``baselines``, ``dataset``).

INTEGRATION.md puts ``semanticlidarunc_amd/`` ahead of the reference's ``src/`` on ``sys.path`` so that e.g.
``from models.evaluator import IoUEvaluator`` resolves here.  Two things must then keep working:

* modules this repo does NOT mirror (``utils.vis_cv2``, ``models.trainer`` ...): every mirrored sub-package extends its
  ``__path__`` with the same-named directories further down ``sys.path`` (``merge_package_path``), so the reference's namespace
  packages still contribute their files;
* names a PARTIALLY mirrored module does not define (``models.probability_helper.build_uncertainty_layers``,
  ``losses.dirichlet_losses._valid_mask`` ...): ``reexport_missing`` loads the shadowed reference file under a private name
  and copies into the mirror every name the mirror lacks.

Both are no-ops when the modules are imported under their ``semanticlidarunc_amd.`` names or when no reference tree is on
``sys.path`` (tests, bench.py, the GPU box)."""
from __future__ import annotations

import importlib.util
import os
import sys
import warnings
from pkgutil import extend_path


def merge_package_path(path, name):
    """``__path__ = merge_package_path(__path__, __name__)`` in a mirrored sub-package's ``__init__``."""
    return extend_path(path, name)


def shadowed_module(module_name: str, module_file: str):
    """The reference module this mirror shadows (loaded once under a private name), or None."""
    if module_name.startswith("semanticlidarunc_amd."):
        return None                                     # imported under the package's own name: nothing is shadowed
    private = "_slu_shadowed_." + module_name
    if private in sys.modules:
        return sys.modules[private]
    rel = os.path.join(*module_name.split(".")) + ".py"
    here = os.path.realpath(module_file)
    for entry in sys.path:
        cand = os.path.join(entry or ".", rel)
        if os.path.isfile(cand) and os.path.realpath(cand) != here:
            try:
                spec = importlib.util.spec_from_file_location(private, cand)
                mod = importlib.util.module_from_spec(spec)
                sys.modules[private] = mod
                spec.loader.exec_module(mod)
                return mod
            except Exception as exc:                    # the reference module needs something this environment lacks
                sys.modules.pop(private, None)
                warnings.warn(f"{module_name}: could not load the shadowed reference module {cand} ({exc!r}); "
                              "names this mirror does not define stay undefined")
                return None
    return None


def reexport_missing(module_name: str, module_file: str, namespace: dict) -> bool:
    """Copy into ``namespace`` the names the shadowed reference module defines and the mirror does not.
    Returns True if a shadowed file was found and loaded."""
    mod = shadowed_module(module_name, module_file)
    if mod is None:
        return False
    for key, value in vars(mod).items():
        if key not in namespace and not (key.startswith("__") and key.endswith("__")):
            namespace[key] = value
    return True
