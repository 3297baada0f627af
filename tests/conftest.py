import os
import sys

import pytest

# PyTorch-ROCm and libamg_hip.so both depend on a libamdhip64.so.7 (torch bundles its
# own); the first one loaded serves the whole process, and torch cannot initialise on
# the system copy.  Tests that hand torch tensors to the C ABI need torch's copy, so it
# is loaded before the library, whatever the test order.
try:
    import torch  # noqa: F401
except ImportError:  # the C-ABI tests do not need it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def amg():
    """ctypes binding of libamg_hip.so (the product's C ABI)."""
    pkg = os.path.join(ROOT, "algebraic-multigrid_amd")
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    import amg_ctypes
    amg_ctypes.lib()
    return amg_ctypes
