"""The C++ drop-in layer (include/amg/*.hpp over the C ABI): the reference's own
test driver, restated in tests/cpp/testlib_amd.cpp, must build here (CPU) and
pass on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "algebraic-multigrid_amd")],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CPP], stdout=subprocess.DEVNULL)
    return os.path.join(CPP, "testlib_amd")


def test_dropin_headers_compile_and_link():
    exe = _build()
    assert os.path.exists(exe)
    # every reference header has its drop-in
    for h in ("common.hpp", "grid.hpp", "interpolator.hpp", "multigrid.hpp", "smoother.hpp"):
        assert os.path.exists(os.path.join(ROOT, "include", "amg", h))


@pytest.mark.gpu
def test_reference_test_driver_passes_on_gpu():
    exe = os.path.join(CPP, "testlib_amd")
    if not os.path.exists(exe):
        exe = _build()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    out = p.stdout
    assert p.returncode == 0, out[-3000:] + p.stderr[-2000:]
    # the lines the reference prints (image/README/output.png)
    assert "SPGS converged after 900 iterations." in out
    assert "AMG converged after 35 iterations." in out
    assert "All tests passed" in out
    sizes = [int(x) for x in out.split("Dofs at Levels in Multigrid:")[1].split()[:8]]
    assert sizes == [1225, 612, 305, 152, 75, 37, 18, 8]
