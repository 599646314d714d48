import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree libpsvo_hip.so; built on demand (hipcc cross-compiles gfx950 without a GPU)."""
    from psvo_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_lib(verbose=False)
    return _lib.LIB_PATH
