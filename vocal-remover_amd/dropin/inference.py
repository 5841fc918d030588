"""Shadow of the reference's top-level `inference` module (Separator only)."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
import __graft_entry__ as _ge  # noqa: E402

_pkg = _ge.load_package()
from vocal_remover_amd.inference import Separator  # noqa: E402,F401
