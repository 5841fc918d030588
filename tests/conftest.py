import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = '/root/reference'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_product():
    """Import the product package (directory name has a hyphen) as ``vocal_remover_amd``."""
    import __graft_entry__
    return __graft_entry__.load_package()


@pytest.fixture(scope='session')
def vr():
    return load_product()


@pytest.fixture(scope='session')
def reference_lib():
    """The reference's own python, importable only in the build container."""
    if not os.path.isdir(os.path.join(REFERENCE, 'lib')):
        pytest.skip('/root/reference not present (GPU box)')
    import types
    for name in ('librosa', 'soundfile', 'cv2'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['cv2'].IMREAD_COLOR = 1     # lib/utils.py:7 reads it in a default argument
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import lib.nets as ref_nets        # noqa: E402  (reference package `lib`)
    import inference as ref_inference  # noqa: E402
    return types.SimpleNamespace(nets=ref_nets, inference=ref_inference)
