"""Shadow of the reference's lib/dataset.py: the MI355X implementations; names this package does not provide fall through to the
checkout's own lib/dataset.py (module __getattr__), e.g. image dumps and the torch layer classes, which are outside the hot path."""
from lib import _passthrough, _pkg  # noqa: F401  (importing `lib` loads vocal_remover_amd)
from vocal_remover_amd.dataset import *  # noqa: E402,F401,F403
import vocal_remover_amd.dataset as _impl  # noqa: E402


def __getattr__(name):
    if name.startswith('__'):
        raise AttributeError(name)
    if hasattr(_impl, name):
        return getattr(_impl, name)
    try:
        return getattr(_passthrough('dataset'), name)
    except ImportError as e:
        raise AttributeError('lib.dataset.%s: not part of the MI355X path and %s' % (name, e))
