"""Host-side code (Newton / More-Thuente driver, 6x6 solves, reducers, SE(3) algebra) under
AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only -- GPU sanitizers are not available on
the pool; the HIP kernels are covered by the parity tests instead."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_asan_ubsan():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-C", d, "sanitize_host"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(d, "sanitize_host")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.strip().endswith("PASS")
