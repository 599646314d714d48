import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The fp64 oracle is PyTorch on the host: a GPU box shows every core of the machine to torch but grants this job a share of
    # them (16 for one GPU), and with one thread per visible core the oracle's passes ran 2 - 4 x slower from box to box.
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree libpsvo_hip.so; built on demand (hipcc cross-compiles gfx950 without a GPU)."""
    from psvo_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_lib(verbose=False)
    return _lib.LIB_PATH
