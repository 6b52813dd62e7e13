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


class _HipMem:
    """Device buffers through the HIP runtime the engine itself links (ctypes on libamdhip64):
    GPU tests that need device pointers use this instead of torch, whose bundled runtime cannot
    initialise once the engine's has."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.live = []

    def upload(self, arr):
        import numpy as np
        a = np.ascontiguousarray(arr)
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), max(a.nbytes, 4)) == 0
        assert self.rt.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0   # hipMemcpyHostToDevice
        self.live.append(p)
        return p.value

    def write(self, ptr, arr):
        """Overwrites an existing device buffer in place (a caller re-filling its scan buffer)."""
        import numpy as np
        a = np.ascontiguousarray(arr)
        assert self.rt.hipMemcpy(self.C.c_void_p(ptr), a.ctypes.data, a.nbytes, 1) == 0

    def free_all(self):
        for p in self.live:
            self.rt.hipFree(p)
        self.live = []


@pytest.fixture()
def hipmem():
    m = _HipMem()
    yield m
    m.free_all()
