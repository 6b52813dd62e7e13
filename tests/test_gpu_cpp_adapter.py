"""Runs the C++ adapter test (the reference's PclOmp convergence test written against
include/ndt_hip/ndt_hip.hpp) on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_adapter_reference_convergence():
    exe = os.path.join(ROOT, "tests", "cpp", "test_adapter")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "PASS" in p.stdout


def test_cpp_adapter_compiles():
    """The PCL-free face of the adapter builds with plain g++ against the C-ABI library."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "test_adapter"))
