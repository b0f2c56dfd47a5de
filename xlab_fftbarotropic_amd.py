"""Importable alias of the package directory `xlab-fftbarotropic_amd/`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("xlab-fftbarotropic_amd")
globals().update({k: getattr(_pkg, k) for k in dir(_pkg) if not k.startswith("__")})
package = _pkg
