import os
import sys

import pytest
import torch  # noqa: F401  -- before libphylign_match.so: torch bundles its own libamdhip64 with the same SONAME as
#                              /opt/rocm's; whichever is loaded first serves both, and torch only finds GPUs with its own

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def pm():
    """The HIP library bound to cuda:0; fails loudly when it is missing."""
    from phylign_amd import _lib
    _lib.load()
    _lib.init(0)
    return _lib
