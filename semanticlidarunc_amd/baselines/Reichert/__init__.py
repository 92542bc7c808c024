# Mirrors one of the reference's import paths; the same-named directories further down sys.path (the reference's namespace
# packages) keep contributing the modules that are not mirrored here (see semanticlidarunc_amd/_shadow.py).
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
