"""The library's HOST setup code (algebraic-multigrid_amd/csrc/host_setup.cpp: Galerkin hierarchy,
layout encoders, coarsest-factor schedules, colourings, strength-based coarsening, the window
generators) under AddressSanitizer + UndefinedBehaviorSanitizer -- CPU build only, no GPU
(the reference runs valgrind memcheck in CI, `.github/workflows/sca.yml:77-79`)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_setup_under_asan_and_ubsan():
    d = os.path.join(ROOT, "tests", "sanitize")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([os.path.join(d, "host_setup_asan")], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "host_setup_asan ok" in p.stdout and "runtime error" not in p.stderr
