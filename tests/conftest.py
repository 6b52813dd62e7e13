import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes binding over the C-ABI)."""
    return ge.load_package()


@pytest.fixture(scope="session")
def O():
    """The CPU oracle -- the checker, never the thing measured."""
    o = ge.load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def S(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
