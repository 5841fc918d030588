"""Shadow of the reference's `lib` package.  With vocal-remover_amd/dropin ahead of the reference checkout on sys.path
(vocal-remover_amd/run.py arranges that; PYTHONPATH alone cannot: the script's directory is sys.path[0]),
`from lib import nets, spec_utils, dataset` resolve to the MI355X path.  Every other submodule (`lib.utils`, `lib.layers`)
passes through to the checkout's own lib/: its directory is appended to this package's __path__."""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_here)))
if _root not in _sys.path:
    _sys.path.append(_root)
import __graft_entry__ as _ge  # noqa: E402

_pkg = _ge.load_package()

for _p in list(_sys.path):
    _cand = _os.path.join(_os.path.abspath(_p or _os.getcwd()), 'lib')
    if _cand != _here and _os.path.isfile(_os.path.join(_cand, '__init__.py')) and _cand not in __path__:
        __path__.append(_cand)           # the reference checkout's lib/: utils.py, layers.py, ...


def _passthrough(name):
    """The checkout's own lib/<name>.py as a private module (for names of a shadowed module that are outside the hot path)."""
    import importlib.util
    key = 'lib._reference_' + name
    if key in _sys.modules:
        return _sys.modules[key]
    for _d in __path__[1:]:
        _f = _os.path.join(_d, name + '.py')
        if _os.path.isfile(_f):
            spec = importlib.util.spec_from_file_location(key, _f)
            mod = importlib.util.module_from_spec(spec)
            _sys.modules[key] = mod
            spec.loader.exec_module(mod)
            return mod
    raise ImportError('no reference checkout next to the running script: lib/%s.py not found' % name)
