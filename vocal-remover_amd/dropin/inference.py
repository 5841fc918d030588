"""Shadow of the reference's top-level `inference` module for scripts that import it (pseudo.py:9 uses inference.Separator)."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.append(_root)
import __graft_entry__ as _ge  # noqa: E402

_pkg = _ge.load_package()
from vocal_remover_amd.inference import Separator, main  # noqa: E402,F401
